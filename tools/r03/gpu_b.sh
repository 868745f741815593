#!/bin/bash
# round 3, call B: the quad-organised packet fetch + child references from LDS: full GPU suite, then A/B on C2 at 256 spp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03b
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03b/gpu_tests.log 2>&1; tail -3 gpurun_out/r03b/gpu_tests.log
{ for i in 1 2; do tools/ab.sh 256 base q0c0 q1c0 q0c1; done; WL=c3 tools/ab.sh 256 base; WL=c5 tools/ab.sh 64 base; } > gpurun_out/r03b/ab.txt 2>&1
cat gpurun_out/r03b/ab.txt
timeout 300 python tools/gpu_stats.py 64 c2 > gpurun_out/r03b/stats_c2.txt 2>&1; tail -8 gpurun_out/r03b/stats_c2.txt
