#!/bin/bash
# round 3, call G: (1) what separates the device from the oracle on the C2 stand-in: shipped build vs IEEE build with an exact triangle test,
# on a room with baked transforms, with instances, and at the bench size; (2) heavy-kernel occupancy A/B; (3) default slice size
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03g
timeout 600 python /dev/stdin < tools/r03/ieee_exp.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03g/ieee_exp.txt
{
echo "== C5 512 spp default slice"; timeout 300 python tools/c5_ablation.py 512 full
for l in hw3 hw4; do echo "== C5 512 spp heavy kernel at ${l#hw} waves"; LRHIP_LIB=$PWD/luisarender_amd/lib/variants/liblrhip_$l.so timeout 300 python tools/c5_ablation.py 512 full; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03g/ab.txt
