#!/bin/bash
# what the byte-texel branch in texel_at costs scenes that keep float texels: prev = the kernels of the commit before it, p0 = today's without the wave priorities, base = today's
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05zh; O=gpurun_out/r05zh
SCHED=auto REPEAT=3 timeout 500 python tools/ab_libs.py 512 c5 prev p0 base prev 2>&1 | grep "^c5" | tee $O/ab_prev.txt
SCHED=auto REPEAT=3 timeout 500 python tools/ab_libs.py 256 c3 prev p0 base prev 2>&1 | grep "^c3" | tee -a $O/ab_prev.txt
