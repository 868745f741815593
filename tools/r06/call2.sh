#!/bin/bash
# round 6, call 2: the stall probe on the three kernels that own the frames + the counter list + thread trace attempt + GPU tests of the refactored node step
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06b; mkdir -p $O
rocprofv3-avail list > $O/counters.txt 2>&1
( cd /tmp && timeout 120 rocprofv3 --att --kernel-trace -d /tmp/att_try -o att -- python $GRAFT_REPO_ROOT/bench.py --workload c1 --spp 4 --steps 1 --warmup 0 --no-cpu-baseline --no-pmc --no-extra --no-stats ) > $O/att_attempt.log 2>&1
echo "rc=$?" >> $O/att_attempt.log
SCHED=auto REPEAT=2 timeout 300 python tools/ab_libs.py 256 c2,c3 base 2>&1 | grep -v amdgpu.ids | tee $O/after_refactor.txt
SCHED=auto REPEAT=2 timeout 300 python tools/ab_libs.py 64 c4 base 2>&1 | grep -v amdgpu.ids | tee -a $O/after_refactor.txt
SCHED=auto REPEAT=2 timeout 300 python tools/ab_libs.py 512 c5 base 2>&1 | grep -v amdgpu.ids | tee -a $O/after_refactor.txt
timeout 300 python tools/stall_probe.py c2 256 $O/stalls_c2.json 2>&1 | grep -v amdgpu.ids | tee $O/stalls_c2.txt
timeout 300 python tools/stall_probe.py c4 64 $O/stalls_c4.json 2>&1 | grep -v amdgpu.ids | tee $O/stalls_c4.txt
timeout 300 python tools/stall_probe.py c5 512 $O/stalls_c5.json 2>&1 | grep -v amdgpu.ids | tee $O/stalls_c5.txt
SAMPLER=PaddedSobol timeout 300 python tools/stall_probe.py c2 256 $O/stalls_c2_sobol.json 2>&1 | grep -v amdgpu.ids | tee $O/stalls_c2_sobol.txt
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/gpu_tests.txt
