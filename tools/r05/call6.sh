#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05f; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; grep -E "passed|failed|FAILED|^E  |Fatal|Error" $O/gpu_tests.log | tail -8
SAMPLER=PaddedSobol SCHED=pool timeout 900 python tools/ab_libs.py 256 c2 r04 base 2>&1 | grep -v amdgpu.ids | tee $O/ab_c2_256_psobol.txt
SAMPLER=Sobol SCHED=pool timeout 900 python tools/ab_libs.py 256 c2 r04 base 2>&1 | grep -v amdgpu.ids | tee $O/ab_c2_256_sobol.txt
SAMPLER=PCG32 SCHED=pool timeout 900 python tools/ab_libs.py 256 c2 r04 base 2>&1 | grep -v amdgpu.ids | tee $O/ab_c2_256_pcg.txt
SCHED=pool timeout 900 python tools/ab_libs.py 256 c2 r04 prev base 2>&1 | grep -v amdgpu.ids | tee $O/ab_c2_256.txt
