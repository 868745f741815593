#!/bin/bash
# packed texels: rows vs 8 x 4 tiles, two-form decode vs one expression (C4 pool kernel 4116, films must stay bit-identical)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05ze; O=gpurun_out/r05ze
SCHED=pool REPEAT=3 timeout 500 python tools/ab_libs.py 64 c4 base lean tile tilelean base 2>&1 | grep "^c4" | tee $O/ab_texel_layout.txt
