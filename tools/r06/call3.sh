#!/bin/bash
# round 6, call 3: the stall probe with one wave in 32 taking the timestamps
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06c; mkdir -p $O
timeout 300 python tools/stall_probe.py c2 256 $O/stalls_c2.json 2>&1 | grep -v amdgpu.ids | tee $O/stalls_c2.txt
timeout 300 python tools/stall_probe.py c4 64 $O/stalls_c4.json 2>&1 | grep -v amdgpu.ids | tee $O/stalls_c4.txt
timeout 300 python tools/stall_probe.py c5 512 $O/stalls_c5.json 2>&1 | grep -v amdgpu.ids | tee $O/stalls_c5.txt
SAMPLER=PaddedSobol timeout 300 python tools/stall_probe.py c2 256 $O/stalls_c2_sobol.json 2>&1 | grep -v amdgpu.ids | tee $O/stalls_c2_sobol.txt
timeout 300 python tools/stall_probe.py c3 256 $O/stalls_c3.json 2>&1 | grep -v amdgpu.ids | tee $O/stalls_c3.txt
