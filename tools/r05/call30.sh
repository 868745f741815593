#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=gpurun_out/r05za; mkdir -p $O
SCHED=pool timeout 900 python tools/ab_libs.py 128 c4,c3 base tci 2>&1 | grep -v amdgpu.ids | tee $O/ab_c4_c3_tci.txt
