#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4
bash tools/ab.sh 256 base w5 w3 rf32 rf48 base
} > gpurun_out/r02u.txt 2>&1
cat gpurun_out/r02u.txt
