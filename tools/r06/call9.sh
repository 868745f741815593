#!/bin/bash
# round 6, call 9: the byte-texel decode compiled out of the lean kernels (nobyte) against the shipped ones; lobe form 2 + xyz-only lookups in the shipped library
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06k; mkdir -p $O
SCHED=auto REPEAT=2 timeout 300 python tools/ab_libs.py 256 c2,c3 base nobyte base 2>&1 | grep -v amdgpu.ids | tee $O/ab.txt
SCHED=auto REPEAT=2 timeout 300 python tools/ab_libs.py 512 c5 base nobyte 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
SCHED=auto REPEAT=2 timeout 300 python tools/ab_libs.py 64 c4 base 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/gpu_tests.txt
