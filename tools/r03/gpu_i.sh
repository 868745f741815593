#!/bin/bash
# round 3, call I: wavefront A/B on C5 at 512 spp: Disney inline in the lean kernel (only Mix / Layered parked), continuation item cap
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03i
{
echo "== base"; timeout 300 python tools/c5_ablation.py 512 full
for l in wfd wi256 wi1024; do echo "== $l"; LRHIP_LIB=$PWD/luisarender_amd/lib/variants/liblrhip_$l.so timeout 300 python tools/c5_ablation.py 512 full; done
echo "== base again"; timeout 300 python tools/c5_ablation.py 512 full
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03i/ab.txt
