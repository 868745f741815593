#!/bin/bash
# Experiment / build helper: compile ONE megakernel variant with a cache-policy modifier (nt | sc1 | "sc0 sc1") on every scratch
# (register spill) instruction -- the compiler has no knob for it -- and write the ordinary host+device object:
#   tools/patch_scratch_policy.sh <mask> <modifier> <out.o> [extra -D / compiler flags...]
set -e
MASK=$1; MOD=$2; OUT=$3; shift 3
L=/opt/rocm/lib/llvm/bin; T=$(mktemp -d)
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-hip-fp32-correctly-rounded-divide-sqrt -fapprox-func -fno-slp-vectorize -DLR_VARIANT=$MASK $*"
SRC=$(dirname $0)/../luisarender_amd/csrc/hip/megapath_variant.hip
/opt/rocm/bin/hipcc $F --cuda-device-only -S -o $T/dev.s $SRC 2>/dev/null
sed -E "/^\s*scratch_(load|store)_/ s/^([^;]*[^; \t])(\s*;.*)?\$/\1 $MOD\2/" $T/dev.s > $T/dev_p.s
$L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $T/dev_p.s -o $T/dev.o
$L/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $T/dev.out $T/dev.o
$L/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$T/dev.out -output=$T/dev.hipfb
/opt/rocm/bin/hipcc $F --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $T/dev.hipfb -c $SRC -o $OUT 2>/dev/null
echo "$OUT: $(grep -c "^\s*scratch_" $T/dev_p.s) scratch instructions with '$MOD'"
rm -rf $T
