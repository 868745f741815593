#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4
echo "== shipped"; timeout 300 python tools/c5_ablation.py 64 full no_layered no_layered_mix basic_no_alpha 2>&1 | grep -v amdgpu.ids
for lib in b8 b12 b24; do echo "== $lib"; LRHIP_LIB=luisarender_amd/lib/variants/liblrhip_$lib.so LRHIP_FORCE_FEATURES=124 timeout 300 python tools/c5_ablation.py 64 full 2>&1 | grep -v amdgpu.ids; done
for lib in m60b8 m60b16; do echo "== $lib"; LRHIP_LIB=luisarender_amd/lib/variants/liblrhip_$lib.so LRHIP_FORCE_FEATURES=60 timeout 300 python tools/c5_ablation.py 64 no_layered 2>&1 | grep -v amdgpu.ids; done
} > gpurun_out/r02w.txt 2>&1
cat gpurun_out/r02w.txt
