#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=gpurun_out/r05t; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; grep -E "passed|failed|FAILED|^E  |Fatal|Error" $O/gpu_tests.log | tail -8
timeout 600 python tools/c5_ablation.py 2048 full alpha_only 2>&1 | grep -v amdgpu | tee $O/c5_2048.txt
SCHED=both timeout 900 python tools/ab_libs.py 256 c2 r04 base 2>&1 | grep -v amdgpu.ids | tee $O/ab_c2_256.txt
