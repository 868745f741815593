#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
for lib in l2 l3nosgpr l3slp l4; do
echo "=== $lib"
LRHIP_LIB=luisarender_amd/lib/variants/liblrhip_$lib.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "layered_closure and not mix" 2>&1 | grep "passed\|failed\|layered parity"
LRHIP_LIB=luisarender_amd/lib/variants/liblrhip_$lib.so LRHIP_FORCE_FEATURES=124 timeout 300 python tools/c5_ablation.py 64 full 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r02v.txt 2>&1
cat gpurun_out/r02v.txt
