#!/bin/bash
# the automatic scheduler rule once more at the final hash (wave priorities moved both kernel families)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05zi
timeout 240 python tools/sched_sweep.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05zi/sched_sweep.txt
