#!/bin/bash
# experiment: the scene-level "some image is packed" flag in bit 0 of the texel table's address; float-only lookups behind the wave-uniform branch (ub) against the shipped per-texture test (base)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05zk; O=gpurun_out/r05zk
SCHED=auto REPEAT=3 timeout 200 python tools/ab_libs.py 256 c5 base ub base 2>&1 | grep "^c5" | tee $O/ab_uniform_branch.txt
SCHED=auto REPEAT=3 timeout 200 python tools/ab_libs.py 256 c3 base ub base 2>&1 | grep "^c3" | tee -a $O/ab_uniform_branch.txt
SCHED=pool REPEAT=3 timeout 200 python tools/ab_libs.py 256 c2 base ub base 2>&1 | grep "^c2" | tee -a $O/ab_uniform_branch.txt
SCHED=auto REPEAT=2 timeout 200 python tools/ab_libs.py 32 c4 base ub 2>&1 | grep "^c4" | tee -a $O/ab_uniform_branch.txt
