#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=gpurun_out/r05l; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "sub_ranges or fixed_point or wavefront" > $O/gpu_tests.log 2>&1; grep -E "passed|failed|FAILED|^E  |Fatal|Error" $O/gpu_tests.log | tail -8
SCHED=pool timeout 900 python tools/ab_libs.py 256 c2 base s48 s62 i8 i28 t12 t12s62 2>&1 | grep -v amdgpu.ids | tee $O/ab_c2_256_thresholds.txt
SCHED=pool timeout 900 python tools/ab_libs.py 64 c3 base s48 s62 i8 i28 t12 t12s62 2>&1 | grep -v amdgpu.ids | tee $O/ab_c3_64_thresholds.txt
