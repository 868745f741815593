#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=gpurun_out/r05j; mkdir -p $O
timeout 900 python tools/sched_sweep.py 2>&1 | grep -v amdgpu | tee $O/sched_sweep.txt
for b in base hw1 hw3 hw4; do
  if [ "$b" = base ]; then unset LRHIP_LIB; else export LRHIP_LIB=$R/luisarender_amd/lib/variants/liblrhip_$b.so; fi
  echo "== $b"; timeout 600 python tools/c5_ablation.py 1024 full 2>&1 | grep -v amdgpu | tee -a $O/c5_heavy_waves.txt
done
