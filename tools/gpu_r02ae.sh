#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4
echo "== shipped (call-making variants with SGPR spills in memory; parking)"; timeout 300 python tools/c5_ablation.py 64 full no_layered no_layered_mix 2>&1 | grep -v amdgpu.ids
for lib in s3 s4b24 s4b12; do echo "== $lib"; LRHIP_LIB=luisarender_amd/lib/variants/liblrhip_$lib.so LRHIP_FORCE_FEATURES=124 timeout 300 python tools/c5_ablation.py 64 full 2>&1 | grep -v amdgpu.ids; done
} > gpurun_out/r02ae.txt 2>&1
cat gpurun_out/r02ae.txt
