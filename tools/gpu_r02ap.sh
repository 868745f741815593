#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
for lib in pk1 pk2 pk3 pk4; do echo "== $lib"; LRHIP_LIB=luisarender_amd/lib/variants/liblrhip_$lib.so timeout 300 python tools/c5_ablation.py 64 full no_layered 2>&1 | grep -v amdgpu.ids; done
} > gpurun_out/r02ap.txt 2>&1
cat gpurun_out/r02ap.txt
