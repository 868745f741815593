#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
for lib in nosgpr nosgpr3; do
echo "=== $lib"
LRHIP_LIB=luisarender_amd/lib/variants/liblrhip_$lib.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "(twins and (layered or mix_alpha)) or (layered_closure and not mix)" 2>&1 | grep "passed\|failed"
LRHIP_LIB=luisarender_amd/lib/variants/liblrhip_$lib.so LRHIP_FORCE_FEATURES=124 timeout 300 python tools/c5_ablation.py 64 full 2>&1 | grep -v amdgpu.ids
LRHIP_HEAVY_QUEUE=0 LRHIP_LIB=luisarender_amd/lib/variants/liblrhip_$lib.so LRHIP_FORCE_FEATURES=124 timeout 300 python tools/c5_ablation.py 64 full 2>&1 | grep -v amdgpu.ids
done
LRHIP_LIB=luisarender_amd/lib/variants/liblrhip_nosgpr.so timeout 300 python tools/c5_ablation.py 64 no_layered 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r02ad.txt 2>&1
cat gpurun_out/r02ad.txt
