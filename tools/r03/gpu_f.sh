#!/bin/bash
# round 3, call F: wavefront mode with adaptive continuation items: full GPU suite, C5 slice-size sweep, per-round timeline
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03f; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -s > gpurun_out/r03f/gpu_tests.log 2>&1; grep -E "passed|failed|C1 full|C2-class|wavefront vs|FAILED|Error|^E  " gpurun_out/r03f/gpu_tests.log | tail -14
{
for sp in 33554432 67108864 134217728; do echo "== C5 512 spp, slice paths $sp"; WF_SLICE_PATHS=$sp timeout 300 python tools/c5_ablation.py 512 full; done
echo "== C5 512 spp all-in-one"; WAVEFRONT=0 timeout 300 python tools/c5_ablation.py 512 full
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03f/ab.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/r03f/trace -o trace -- python $R/tools/c5_ablation.py 64 full > $R/gpurun_out/r03f/trace.log 2>&1 )
python tools/wf_trace.py gpurun_out/r03f/trace | head -12 | tee gpurun_out/r03f/wf_trace.txt
