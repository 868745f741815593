#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
for lib in tc tc4 tc3; do
  echo "== $lib"
  LRHIP_LIB=luisarender_amd/lib/variants/liblrhip_$lib.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_ref_golden.py -m gpu -x -q -k "mix or layered or kitchen or c5" 2>&1 | tail -3
  LRHIP_LIB=luisarender_amd/lib/variants/liblrhip_$lib.so timeout 300 python tools/c5_ablation.py 64 full no_layered no_layered_mix
done
echo "== shipped"; timeout 300 python tools/c5_ablation.py 64 full no_layered no_layered_mix
} > gpurun_out/r02j.txt 2>&1
cat gpurun_out/r02j.txt
