#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=gpurun_out/r05w; mkdir -p $O
SCHED=pool timeout 900 python tools/ab_libs.py 128 c4,c3 base w3d 2>&1 | grep -v amdgpu.ids | tee $O/ab_c4_c3_w3.txt
