#!/bin/bash
# LR_POOL_TURNOVER_LANES again under the wave priorities (12 shipped; 8 / 16 / 24)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05zg; O=gpurun_out/r05zg
SCHED=pool REPEAT=3 timeout 500 python tools/ab_libs.py 512 c2 base t8 t16 t24 base 2>&1 | grep "^c2" | tee $O/ab_turnover.txt
SCHED=pool REPEAT=3 timeout 500 python tools/ab_libs.py 256 c3 base t8 t16 t24 base 2>&1 | grep "^c3" | tee -a $O/ab_turnover.txt
