#!/bin/bash
# s_setprio: the traversal loop (ptr) or the shading block (psh) at priority 3, the other at 0 -- pool kernels 4096 (C2) and 4100 (C3)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05zf; O=gpurun_out/r05zf
SCHED=pool REPEAT=3 timeout 500 python tools/ab_libs.py 512 c2 base ptr psh base 2>&1 | grep "^c2" | tee $O/ab_setprio.txt
SCHED=pool REPEAT=3 timeout 500 python tools/ab_libs.py 256 c3 base ptr psh base 2>&1 | grep "^c3" | tee -a $O/ab_setprio.txt
