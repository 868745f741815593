#!/bin/bash
# round 3, call K: split of the shading block (new section counters) on C2 / C3 / C5
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03k
for w in c2 c3; do timeout 300 python tools/gpu_stats.py 64 $w 2>&1 | grep -v amdgpu.ids | tail -9 | tee gpurun_out/r03k/stats_$w.txt; done
timeout 300 python tools/gpu_stats.py 256 c2 2>&1 | grep -v amdgpu.ids | tail -9 | tee gpurun_out/r03k/stats_c2_256.txt
