#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05z; mkdir -p $O
timeout 600 python tools/c4_ablation.py 128 2>&1 | grep -v amdgpu | tee $O/c4_ablation.txt
