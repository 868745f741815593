#!/bin/bash
# round 6, call 1: is PC sampling available on the box (stochastic, else host trap)?  + this box's baseline of every configuration
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06a; mkdir -p $O
rocprofv3-avail info --pc-sampling > $O/pcs_info.txt 2>&1
tools/pc_sample.sh r06a c2 256 stochastic 1048576
if ! grep -q '"samples": [1-9]' $O/pcs_c2.json 2>/dev/null; then
  cp $O/pcs_c2.log $O/pcs_c2_stochastic_failed.log
  tools/pc_sample.sh r06a c2 256 host_trap 50
fi
SCHED=auto REPEAT=2 timeout 300 python tools/ab_libs.py 256 c2,c3 base 2>&1 | grep -v amdgpu.ids | tee $O/baseline.txt
SCHED=auto REPEAT=2 timeout 300 python tools/ab_libs.py 64 c4 base 2>&1 | grep -v amdgpu.ids | tee -a $O/baseline.txt
SCHED=auto REPEAT=2 timeout 300 python tools/ab_libs.py 512 c5 base 2>&1 | grep -v amdgpu.ids | tee -a $O/baseline.txt
SAMPLER=PaddedSobol SCHED=auto REPEAT=2 timeout 300 python tools/ab_libs.py 256 c2 base 2>&1 | grep -v amdgpu.ids | tee -a $O/baseline.txt
