#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=gpurun_out/r05i; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; grep -E "passed|failed|FAILED|^E  |Fatal|Error" $O/gpu_tests.log | tail -8
for e in "X=1" "PARK_DISNEY=1" "WF_SLICE_PATHS=536870912" "WF_SLICE_PATHS=134217728"; do
  echo "== $e"; env $e timeout 600 python tools/c5_ablation.py 2048 full 2>&1 | grep -v amdgpu | tee -a $O/c5_ab_2048.txt
done
