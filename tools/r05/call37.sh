#!/bin/bash
# s_setprio, third pass: shading 3 / 1 with the fetch at 3 (p33, p13), and the whole chain from the sorted children to the next fetch at 3 (pch), against pfe (shading 2, fetch 3)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05zf; O=gpurun_out/r05zf
SCHED=pool REPEAT=3 timeout 500 python tools/ab_libs.py 512 c2 base pfe p33 p13 pch base 2>&1 | grep "^c2" | tee $O/ab_setprio3.txt
SCHED=pool REPEAT=3 timeout 500 python tools/ab_libs.py 256 c3 base pfe p33 p13 pch base 2>&1 | grep "^c3" | tee -a $O/ab_setprio3.txt
