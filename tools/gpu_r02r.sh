#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
for lib in inl rf32 rf48; do echo "== $lib"; LRHIP_LIB=luisarender_amd/lib/variants/liblrhip_$lib.so timeout 300 python tools/c5_ablation.py 64 full no_layered; done
} 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r02r.txt
cat gpurun_out/r02r.txt
