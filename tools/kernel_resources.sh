#!/bin/bash
# Compile the given megakernel variants (feature masks) and print registers / spills / scratch / code size:
#   tools/kernel_resources.sh [-D...] 0 4 20 60 124
DEFS=""; MASKS=""
for a in "$@"; do case $a in -D*) DEFS="$DEFS $a";; *) MASKS="$MASKS $a";; esac; done
OUT=/tmp/kres; mkdir -p $OUT; L=/opt/rocm/lib/llvm/bin
for m in $MASKS; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -fno-hip-fp32-correctly-rounded-divide-sqrt -fapprox-func -fno-slp-vectorize $DEFS \
      -DLR_VARIANT=$m -c -o $OUT/v$m.o luisarender_amd/csrc/hip/megapath_variant.hip 2> $OUT/v$m.err || cat $OUT/v$m.err | head -30
    $L/llvm-objcopy --dump-section .hip_fatbin=$OUT/f$m.bin $OUT/v$m.o && $L/clang-offload-bundler --type=o --input=$OUT/f$m.bin --unbundle \
      --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$OUT/v$m.co 2>/dev/null ) &
done
wait
for m in $MASKS; do
  echo "mask $m: $($L/llvm-readelf --notes $OUT/v$m.co | grep -E "\.vgpr_count|private_segment_fixed|vgpr_spill|sgpr_spill" | tr -s ' ' | tr '\n' ' ') code $(stat -c %s $OUT/v$m.co) B, functions: $($L/llvm-objdump -t $OUT/v$m.co | grep -c 'F .text')"
done
