#!/bin/bash
# round 6, call 10: the byte-texel variant bit (lean kernels without the decode; <12308> for the camera class) -- every configuration + GPU tests
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06l; mkdir -p $O
SCHED=auto REPEAT=2 timeout 300 python tools/ab_libs.py 256 c2,c3 base 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
SCHED=auto REPEAT=2 timeout 300 python tools/ab_libs.py 64 c4 base 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
SCHED=auto REPEAT=2 timeout 300 python tools/ab_libs.py 512 c5 base 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
SCHED=auto REPEAT=2 timeout 300 python tools/ab_libs.py 64 c1 base 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
SAMPLER=PaddedSobol SCHED=auto REPEAT=2 timeout 300 python tools/ab_libs.py 256 c2 base 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/gpu_tests.txt
