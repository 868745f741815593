#!/bin/bash
# round 3, call M: wave-level stack overflow test + slot-as-byte-offset keys: parity quick check, then A/B on C2 at 256 spp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03m
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_ref_golden.py -m gpu -q -x > gpurun_out/r03m/gpu_tests.log 2>&1; grep -E "passed|failed|FAILED|^E  " gpurun_out/r03m/gpu_tests.log | tail -6
{ for i in 1 2; do tools/ab.sh 256 base sf0 sf1k0 sf0k1; done; } 2>&1 | tee gpurun_out/r03m/ab_c2.txt
