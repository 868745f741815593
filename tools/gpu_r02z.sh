#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
echo "== shipped"; timeout 300 python tools/c5_ablation.py 64 full no_layered no_layered_mix 2>&1 | grep -v amdgpu.ids
echo "== inl (texture / env code inlined in the heavy variants)"; LRHIP_LIB=luisarender_amd/lib/variants/liblrhip_inl.so timeout 300 python tools/c5_ablation.py 64 full no_layered no_layered_mix 2>&1 | grep -v amdgpu.ids
echo "== hinl (heavy closures inlined, <60>)"; LRHIP_LIB=luisarender_amd/lib/variants/liblrhip_hinl.so timeout 300 python tools/c5_ablation.py 64 no_layered no_layered_mix 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r02z.txt 2>&1
cat gpurun_out/r02z.txt
