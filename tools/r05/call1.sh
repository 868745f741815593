#!/bin/bash
# round 5, GPU call 1: the new traversal-loop bookkeeping against the round-4 build on the same box (films must be bit-identical), the GPU suite, candidates
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; grep -E "passed|failed|FAILED|^E  |Fatal|Error" $O/gpu_tests.log | tail -8
timeout 600 python tools/ab_libs.py 1024 c2 r04 base 2>&1 | grep -v amdgpu.ids | tee $O/ab_c2_1024.txt
SCHED=pool timeout 600 python tools/ab_libs.py 256 c2 r04 base ps3 lean t4 t12 2>&1 | grep -v amdgpu.ids | tee $O/ab_c2_256_candidates.txt
timeout 600 python tools/ab_libs.py 64 c1,c3,c4 r04 base 2>&1 | grep -v amdgpu.ids | tee $O/ab_others_64.txt
SAMPLER=PaddedSobol SCHED=pool timeout 600 python tools/ab_libs.py 256 c2 r04 base 2>&1 | grep -v amdgpu.ids | tee $O/ab_c2_sobol.txt
LRHIP_SCHEDULER=pool timeout 300 python tools/gpu_stats.py 64 c2 2>&1 | grep -v amdgpu | tail -9 | tee $O/stats_pool_c2.txt
