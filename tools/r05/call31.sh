#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05zb; mkdir -p $O
for t in 2048 1024 512 128; do echo "== texture size $t"; TEXTURE_SIZE=$t timeout 600 python tools/c4_ablation.py 128 full 2>&1 | grep -v amdgpu | tee -a $O/c4_texture_size.txt; done
