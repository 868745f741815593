#!/bin/bash
# the packed-texel lookup out of line (base) against inlined at every use (inl) and the kernels before byte texels, without wave priorities (prev)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05zh; O=gpurun_out/r05zh
SCHED=auto REPEAT=3 timeout 500 python tools/ab_libs.py 512 c5 prev inl base inl 2>&1 | grep "^c5" | tee $O/ab_outofline.txt
SCHED=auto REPEAT=3 timeout 500 python tools/ab_libs.py 256 c3 prev inl base inl 2>&1 | grep "^c3" | tee -a $O/ab_outofline.txt
SCHED=auto REPEAT=3 timeout 500 python tools/ab_libs.py 64 c4 inl base inl 2>&1 | grep "^c4" | tee -a $O/ab_outofline.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "byte_texels or jpeg_texture or textured_disney or address_modes" 2>&1 | tail -2 | tee $O/test.log
