#!/bin/bash
# round 3, call L: per-kind heavy kernels: GPU suite, C5 at 512 spp + timeline
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03l; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -s > gpurun_out/r03l/gpu_tests.log 2>&1; grep -E "passed|failed|wavefront vs|FAILED|^E  " gpurun_out/r03l/gpu_tests.log | tail -10
{ timeout 300 python tools/c5_ablation.py 512 full no_layered no_layered_mix; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03l/c5.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/r03l/trace_c5 -o trace -- python $R/tools/c5_ablation.py 512 full > $R/gpurun_out/r03l/trace_c5.log 2>&1 )
python tools/wf_trace.py gpurun_out/r03l/trace_c5 | head -16 | tee gpurun_out/r03l/wf_trace_c5.txt
