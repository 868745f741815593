#!/bin/bash
# round 3, call J: re-sweep of the lean kernel's parameters on the round-3 node step (C2 at 256 spp), C5 heavy-kernel split by closure family
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03j
{ for i in 1 2; do tools/ab.sh 256 base rf36 rf44 rf48 w5 st12; done; } 2>&1 | tee gpurun_out/r03j/ab_c2.txt
{ timeout 600 python tools/c5_ablation.py 256 full no_layered no_layered_mix no_layered_mix_disney; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03j/c5_split.txt
