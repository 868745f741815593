#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4
timeout 300 python tools/c5_ablation.py 64 2>&1 | grep -v amdgpu.ids
timeout 600 python bench.py --workload c5 --no-pmc --no-extra --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
} > gpurun_out/r02q.txt 2>&1
cat gpurun_out/r02q.txt
