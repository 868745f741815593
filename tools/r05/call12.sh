#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=gpurun_out/r05k; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; grep -E "passed|failed|FAILED|^E  |Fatal|Error" $O/gpu_tests.log | tail -8
timeout 900 python tools/sched_sweep.py quick 2>&1 | grep -v amdgpu | tee $O/sched_sweep_quick.txt
timeout 600 python tools/c5_ablation.py 2048 full 2>&1 | grep -v amdgpu | tee -a $O/c5_2048.txt
