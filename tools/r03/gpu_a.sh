#!/bin/bash
# round 3, call A: VALU / LDS issue-cost table + occupancy diagnostics of the shipped kernels on C2 / C5
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03a
hipcc --offload-arch=gfx950 -O3 tools/valu_peak2.hip -o gpurun_out/r03a/valu_peak2 2> gpurun_out/r03a/valu_build.err
timeout 300 gpurun_out/r03a/valu_peak2 > gpurun_out/r03a/valu_peak.json 2> gpurun_out/r03a/valu_peak.err
rm -f gpurun_out/r03a/valu_peak2
timeout 300 python tools/gpu_stats.py 64 c2 > gpurun_out/r03a/stats_c2.txt 2>&1
timeout 300 python tools/gpu_stats.py 64 c5 > gpurun_out/r03a/stats_c5.txt 2>&1
tail -3 gpurun_out/r03a/stats_c2.txt; head -c 1500 gpurun_out/r03a/valu_peak.json
