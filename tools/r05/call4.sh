#!/bin/bash
# round 5, GPU call 4: the pipelined iteration (LDS reads batched, triangle loads before the pushes, speculative pop) against the serial form
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05d; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; grep -E "passed|failed|FAILED|^E  |Fatal|Error" $O/gpu_tests.log | tail -8
SCHED=both timeout 900 python tools/ab_libs.py 256 c2 r04 base serial nolean t12 t6 2>&1 | grep -v amdgpu.ids | tee $O/ab_c2_256.txt
SCHED=both timeout 300 python tools/ab_libs.py 64 c1 r04 base serial 2>&1 | grep -v amdgpu.ids | tee $O/ab_c1.txt
SCHED=pool timeout 600 python tools/ab_libs.py 1024 c2 r04 base 2>&1 | grep -v amdgpu.ids | tee $O/ab_c2_1024.txt
