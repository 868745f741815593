#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
echo "== shipped"; timeout 300 python tools/c5_ablation.py 64 full no_layered no_layered_mix_disney 2>&1 | grep -v amdgpu.ids
echo "== inl"; LRHIP_LIB=luisarender_amd/lib/variants/liblrhip_inl.so timeout 300 python tools/c5_ablation.py 64 full no_layered 2>&1 | grep -v amdgpu.ids
echo "== 3 waves for Mix and Layered variants"; LRHIP_LIB=luisarender_amd/lib/variants/liblrhip_w3safe.so timeout 300 python tools/c5_ablation.py 64 full no_layered 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r02al.txt 2>&1
cat gpurun_out/r02al.txt
