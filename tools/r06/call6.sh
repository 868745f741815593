#!/bin/bash
# round 6, call 6: closure resolution with the constant slots in the surface record and the looked-up slots in wave-level passes
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06h; mkdir -p $O
SCHED=auto REPEAT=2 timeout 300 python tools/ab_libs.py 256 c2,c3 base 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
SCHED=auto REPEAT=2 timeout 300 python tools/ab_libs.py 64 c4 base 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
SCHED=auto REPEAT=2 timeout 300 python tools/ab_libs.py 512 c5 base 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
SCHED=auto REPEAT=2 timeout 300 python tools/ab_libs.py 64 c1 base 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
