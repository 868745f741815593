#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=gpurun_out/r05v; mkdir -p $O
SCHED=pool timeout 900 python tools/ab_libs.py 256 c2 cur nt1 nt3 2>&1 | grep -v amdgpu.ids | tee $O/ab_c2_256.txt
