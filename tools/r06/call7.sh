#!/bin/bash
# round 6, call 7: looked-up slots in wave-level passes (lobe2) / texels touched ahead (lobe3) against the shipped form
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06i; mkdir -p $O
SCHED=auto REPEAT=2 timeout 300 python tools/ab_libs.py 64 c4 base lobe2 lobe3 base 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
SCHED=auto REPEAT=2 timeout 300 python tools/ab_libs.py 512 c5 base lobe2 lobe3 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
SCHED=auto REPEAT=2 timeout 300 python tools/ab_libs.py 256 c2 base lobe2 lobe3 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
