#!/bin/bash
# A/B: (1) cost of the heavy variants on hits that do not need them; (2) texture/env code inlined in the heavy variants
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
echo "== basic scene on its own variant / forced <60> / forced <124>"
timeout 300 python tools/c5_ablation.py 64 basic_no_alpha
LRHIP_FORCE_FEATURES=60 timeout 300 python tools/c5_ablation.py 64 basic_no_alpha
LRHIP_FORCE_FEATURES=124 timeout 300 python tools/c5_ablation.py 64 basic_no_alpha
echo "== heavy variants with inlined texture/env (liblrhip_inl)"
LRHIP_LIB=luisarender_amd/lib/variants/liblrhip_inl.so LRHIP_FORCE_FEATURES=60 timeout 300 python tools/c5_ablation.py 64 basic_no_alpha
LRHIP_LIB=luisarender_amd/lib/variants/liblrhip_inl.so timeout 300 python tools/c5_ablation.py 64 full no_layered no_layered_mix
echo "== shipped"
timeout 300 python tools/c5_ablation.py 64 full no_layered no_layered_mix
} > gpurun_out/r02g.txt 2>&1
tail -30 gpurun_out/r02g.txt
