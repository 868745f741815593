#!/bin/bash
# heavy-hit parking: parity first, then the batch-size sweep and <60> at 2 / 3 waves per SIMD
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_ref_golden.py -m gpu -x -q -k "mix or layered or disney or kitchen or c5 or golden or twin or variant" 2>&1 | tail -5
for lib in "" hb1 hb12 hb40; do
  echo "== batch ${lib:-24 (shipped)}"
  LRHIP_LIB=${lib:+luisarender_amd/lib/variants/liblrhip_$lib.so} timeout 300 python tools/c5_ablation.py 64 full no_layered no_layered_mix no_layered_mix_disney
done
for lib in mix2 mix3; do
  echo "== $lib (no parking)"
  LRHIP_LIB=luisarender_amd/lib/variants/liblrhip_$lib.so timeout 300 python tools/c5_ablation.py 64 no_layered no_layered_mix no_layered_mix_disney
done
} > gpurun_out/r02h.txt 2>&1
tail -40 gpurun_out/r02h.txt
