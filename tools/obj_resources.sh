#!/bin/bash
# registers / spills / scratch of already-built variant objects: tools/obj_resources.sh <obj dir> <mask>...   (heavy kernels: h<mask>)
D=$1; shift; L=/opt/rocm/lib/llvm/bin; T=/tmp/objres; mkdir -p $T
for m in "$@"; do
  case $m in h*) f=$D/heavy_${m#h}.o;; *) f=$D/variant_$m.o;; esac
  $L/llvm-objcopy --dump-section .hip_fatbin=$T/f.bin $f && $L/clang-offload-bundler --type=o --input=$T/f.bin --unbundle --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/$m.co 2>/dev/null
  echo "$m: $($L/llvm-readelf --notes $T/$m.co | grep -E "\.sgpr_count|\.vgpr_count|private_segment_fixed|vgpr_spill|sgpr_spill|group_segment_fixed" | tr -s ' ' | tr '\n' ' ') code $(stat -c %s $T/$m.co) B"
done
