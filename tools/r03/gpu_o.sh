#!/bin/bash
# round 3, call O: tapered work items: GPU suite, shard probe on C2 (tapered vs uniform), C2 at 256 / 1024 spp, C5
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03o
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r03o/gpu_tests.log 2>&1; grep -E "passed|failed|FAILED|^E  " gpurun_out/r03o/gpu_tests.log | tail -6
timeout 600 python tools/shard_probe.py 1024 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03o/shard_probe.txt
tools/ab.sh 256 base 2>&1 | tee gpurun_out/r03o/c2_256.txt
tools/ab.sh 64 base 2>&1 | tee -a gpurun_out/r03o/c2_256.txt
timeout 300 python tools/c5_ablation.py 512 full 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03o/c5.txt
