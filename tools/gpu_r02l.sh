#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_ref_golden.py -m gpu -x -q -k "mix or layered or disney or kitchen or c5 or golden or twin or variant" 2>&1 | tail -5
timeout 300 python tools/c5_ablation.py 64 full no_layered no_layered_mix
} > gpurun_out/r02l.txt 2>&1
cat gpurun_out/r02l.txt
