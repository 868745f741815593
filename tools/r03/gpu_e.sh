#!/bin/bash
# round 3, call E: where does a wavefront round go?  kernel trace of C5 at 64 spp (default slices), with and without a raised scratch limit
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03e; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/r03e/trace -o trace -- python $R/tools/c5_ablation.py 64 full > $R/gpurun_out/r03e/trace.log 2>&1 )
python tools/wf_trace.py gpurun_out/r03e/trace | tee gpurun_out/r03e/wf_trace.txt
( cd /tmp && HSA_SCRATCH_SINGLE_LIMIT=4294967295 timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/r03e/trace_lim -o trace -- python $R/tools/c5_ablation.py 64 full > $R/gpurun_out/r03e/trace_lim.log 2>&1 )
python tools/wf_trace.py gpurun_out/r03e/trace_lim | tee gpurun_out/r03e/wf_trace_lim.txt
grep full gpurun_out/r03e/trace.log gpurun_out/r03e/trace_lim.log
HSA_SCRATCH_SINGLE_LIMIT=4294967295 timeout 300 python tools/c5_ablation.py 256 full | grep full
WF_SLICE_PATHS=4194304 HSA_SCRATCH_SINGLE_LIMIT=4294967295 timeout 300 python tools/c5_ablation.py 256 full | grep full
rm -rf gpurun_out/r03e/trace/*/*.db.bak 2>/dev/null; du -sh gpurun_out/r03e
