#!/bin/bash
# s_setprio, fourth pass: where the high-priority chain starts (pch: behind the sorted children; pc2: before the sort network; pc3: pch + the leaf's pop) and shading at 1 under pch (pc1s1)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05zf; O=gpurun_out/r05zf
SCHED=pool REPEAT=3 timeout 500 python tools/ab_libs.py 512 c2 base pch pc2 pc3 pc1s1 base 2>&1 | grep "^c2" | tee $O/ab_setprio4.txt
SCHED=pool REPEAT=3 timeout 500 python tools/ab_libs.py 256 c3 base pch pc2 pc3 pc1s1 base 2>&1 | grep "^c3" | tee -a $O/ab_setprio4.txt
