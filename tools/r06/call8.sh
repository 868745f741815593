#!/bin/bash
# round 6, call 8: C5 -- the ALPHA pool kernels under the fused flow, probed; the frame's kernel timeline
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06j; mkdir -p $O
SCHED=auto REPEAT=2 timeout 300 python tools/ab_libs.py 512 c5 base fa 2>&1 | grep -v amdgpu.ids | tee $O/ab_fused_alpha.txt
LIB=probe timeout 300 python tools/stall_probe.py c5 512 $O/stalls1_c5_serial.json 2>&1 | grep -v amdgpu.ids | tee $O/stalls1_c5_serial.txt
LIB=probefa timeout 300 python tools/stall_probe.py c5 512 $O/stalls1_c5_fused.json 2>&1 | grep -v amdgpu.ids | tee $O/stalls1_c5_fused.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d /tmp/wf -o trace -- python $GRAFT_REPO_ROOT/bench.py --workload c5 --spp 2048 --steps 1 --warmup 0 --no-cpu-baseline --no-pmc --no-extra --no-stats > /tmp/wf.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/wf_trace.py /tmp/wf | tee $O/wf_trace_c5_2048spp.txt
