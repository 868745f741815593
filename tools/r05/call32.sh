#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=gpurun_out/r05zc; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; grep -E "passed|failed|FAILED|^E  |Fatal|Error" $O/gpu_tests.log | tail -8
for e in 1 0; do echo "== LRHIP_BYTE_TEXTURES=$e"; LRHIP_BYTE_TEXTURES=$e timeout 600 python tools/c4_ablation.py 128 full 2>&1 | grep -v amdgpu | tee -a $O/byte_textures.txt; LRHIP_BYTE_TEXTURES=$e timeout 600 python tools/c5_ablation.py 512 full 2>&1 | grep -v amdgpu | tee -a $O/byte_textures.txt; done
