#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03s
for l in unsafe ipra0; do echo "== $l"; timeout 900 python tools/sgpr_spill_repro.py luisarender_amd/lib/variants/liblrhip_$l.so 2>&1 | grep -v amdgpu.ids | grep -E "layered|broken"; done | tee gpurun_out/r03s/sgpr_spill_repro.txt
for l in "" ipra0; do echo "== C5 all-in-one <124>, ${l:-shipped (spills to memory)}"; LRHIP_LIB=${l:+$PWD/luisarender_amd/lib/variants/liblrhip_$l.so} WAVEFRONT=0 timeout 300 python tools/c5_ablation.py 256 full 2>&1 | grep full; done | tee -a gpurun_out/r03s/sgpr_spill_repro.txt
