#!/bin/bash
# s_setprio, second pass: the fetch requests of an iteration at priority 3 (pfe: shading 2, tests 0; pfe2: shading 2, tests 1; pfe3: fetch only) against psh (shading 3)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05zf; O=gpurun_out/r05zf
SCHED=pool REPEAT=3 timeout 500 python tools/ab_libs.py 512 c2 base psh pfe pfe2 pfe3 base 2>&1 | grep "^c2" | tee $O/ab_setprio2.txt
SCHED=pool REPEAT=3 timeout 500 python tools/ab_libs.py 256 c3 base psh pfe pfe2 pfe3 base 2>&1 | grep "^c3" | tee -a $O/ab_setprio2.txt
