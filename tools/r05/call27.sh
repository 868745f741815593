#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05x; mkdir -p $O
for w in c2 c3 c4; do echo "== $w"; timeout 300 python tools/gpu_stats.py 64 $w 2>&1 | grep -v amdgpu | tail -8 | tee -a $O/stats_c2_c3_c4.txt; done
