#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
for lib in m3hb6 m3hb8 m3hb12 m3hb16; do
  echo "== $lib"
  LRHIP_LIB=luisarender_amd/lib/variants/liblrhip_$lib.so timeout 300 python tools/c5_ablation.py 64 full no_layered no_layered_mix
done
} > gpurun_out/r02i.txt 2>&1
cat gpurun_out/r02i.txt
