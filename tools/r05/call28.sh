#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=gpurun_out/r05y; mkdir -p $O
SCHED=auto timeout 900 python tools/ab_libs.py 128 c4 base wfd 2>&1 | grep -v amdgpu.ids | tee $O/ab_c4_wf_disney.txt
