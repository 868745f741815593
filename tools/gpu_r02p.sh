#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
echo "== shipped (Layered at 3 waves, batch 12)"; timeout 300 python tools/c5_ablation.py 64 full
for lib in l4 l3b8 l3b16 l3b24; do echo "== $lib"; LRHIP_LIB=luisarender_amd/lib/variants/liblrhip_$lib.so timeout 300 python tools/c5_ablation.py 64 full; done
} 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r02p.txt
cat gpurun_out/r02p.txt
