#!/bin/bash
# wave priorities as built (base = LR_WAVE_PRIORITIES 1) against none (p0) and with the serial flow's leaf fetch raised as well (p2): both schedulers, C1 / C2 / C3 / C5
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05zf; O=gpurun_out/r05zf
SCHED=both REPEAT=3 timeout 500 python tools/ab_libs.py 256 c2 p0 base p2 p0 2>&1 | grep "^c2" | tee $O/ab_setprio5.txt
SCHED=auto REPEAT=3 timeout 500 python tools/ab_libs.py 1024 c1 p0 base p2 p0 2>&1 | grep "^c1" | tee -a $O/ab_setprio5.txt
SCHED=auto REPEAT=3 timeout 500 python tools/ab_libs.py 256 c3 p0 base p2 p0 2>&1 | grep "^c3" | tee -a $O/ab_setprio5.txt
SCHED=auto REPEAT=3 timeout 500 python tools/ab_libs.py 256 c5 p0 base p2 p0 2>&1 | grep "^c5" | tee -a $O/ab_setprio5.txt
