#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_ref_golden.py -m gpu -x -q -k "mix or layered or disney or kitchen or c5 or golden or twin or variant or nested" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -5
echo "== queue (default)"; timeout 300 python tools/c5_ablation.py 64 full no_layered no_layered_mix 2>&1 | grep -v amdgpu.ids
echo "== parking only"; LRHIP_HEAVY_QUEUE=0 timeout 300 python tools/c5_ablation.py 64 full no_layered 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r02ab.txt 2>&1
cat gpurun_out/r02ab.txt
