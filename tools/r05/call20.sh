#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=gpurun_out/r05s; mkdir -p $O
SCHED=both timeout 900 python tools/ab_libs.py 256 c2 base nw fp nwfp 2>&1 | grep -v amdgpu.ids | tee $O/ab_c2_256.txt
SCHED=both timeout 900 python tools/ab_libs.py 64 c1 base nw fp nwfp 2>&1 | grep -v amdgpu.ids | tee $O/ab_c1.txt
