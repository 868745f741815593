#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
LRHIP_LIB=luisarender_amd/lib/variants/liblrhip_lay3.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_ref_golden.py -m gpu -q -k "layered or kitchen or c5 or nested or twin or golden or mix" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -15
} > gpurun_out/r02o.txt 2>&1
cat gpurun_out/r02o.txt
