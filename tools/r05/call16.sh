#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=gpurun_out/r05o; mkdir -p $O
SCHED=pool timeout 900 python tools/ab_libs.py 1024 c2 base pf 2>&1 | grep -v amdgpu.ids | tee $O/ab_c2_1024.txt
SCHED=pool timeout 900 python tools/ab_libs.py 256 c3,c4 base pf 2>&1 | grep -v amdgpu.ids | tee $O/ab_c3_c4_256.txt
