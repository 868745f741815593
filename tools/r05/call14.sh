#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=gpurun_out/r05m; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; grep -E "passed|failed|FAILED|^E  |Fatal|Error" $O/gpu_tests.log | tail -8
for b in a1 a4 base a16 a32; do
  if [ "$b" = base ]; then unset LRHIP_LIB; else export LRHIP_LIB=$R/luisarender_amd/lib/variants/liblrhip_$b.so; fi
  echo "== $b"; timeout 600 python tools/c5_ablation.py 1024 full alpha_only 2>&1 | grep -v amdgpu | tee -a $O/c5_alpha_batch.txt
done
