#!/bin/bash
# compile one megakernel variant with extra -D flags and print its traversal loop's census (tools/isa_census.py) + spills:
#   tools/loop_census.sh <mask> [-D...]
M=$1; shift; T=/tmp/loopc; mkdir -p $T; L=/opt/rocm/lib/llvm/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -fno-hip-fp32-correctly-rounded-divide-sqrt -fapprox-func -fno-slp-vectorize -mllvm -amdgpu-spill-sgpr-to-vgpr=0 "$@" \
  -DLR_VARIANT=$M -c -o $T/v.o luisarender_amd/csrc/hip/megapath_variant.hip 2> $T/err.txt || { head -20 $T/err.txt; exit 1; }
$L/llvm-objcopy --dump-section .hip_fatbin=$T/f.bin $T/v.o && $L/clang-offload-bundler --type=o --input=$T/f.bin --unbundle --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/v.co 2>/dev/null
$L/llvm-objdump -d $T/v.co > $T/v.s
echo "$M $*: $($L/llvm-readelf --notes $T/v.co | grep -E 'vgpr_spill|sgpr_spill|private_segment_fixed' | tr -s ' ' | tr '\n' ' ')"
python tools/isa_census.py $T/v.s | grep -A1 'traversal loop' | cut -c1-330
