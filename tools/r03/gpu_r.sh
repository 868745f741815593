#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03r
timeout 900 python tools/sgpr_spill_repro.py luisarender_amd/lib/variants/liblrhip_unsafe.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03r/sgpr_spill_repro.txt
