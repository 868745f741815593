#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=gpurun_out/r05q; mkdir -p $O
for b in base nf; do
  if [ "$b" = base ]; then unset LRHIP_LIB; else export LRHIP_LIB=$R/luisarender_amd/lib/variants/liblrhip_$b.so; fi
  echo "== $b"; timeout 600 python tools/c5_ablation.py 1024 full alpha_only heavy_no_alpha 2>&1 | grep -v amdgpu | tee -a $O/c5_fused_ab.txt
done
