#!/usr/bin/env python3
"""Static VALU instruction count of a kernel by SOURCE LINE (inlined code is attributed to the line it came from), from a build with
-gline-tables-only: where the code of a variant comes from -- cold paths inlined at many sites show up at once (round 3: the integer
division of the texel wrap and libm's powf in the gamma decode were 3200 of <0>'s 9100 VALU instructions).
    tools/isa_lines.py <mask> [top]"""
import collections
import os
import re
import subprocess
import sys

mask, top = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out, llvm = "/tmp/kres", "/opt/rocm/lib/llvm/bin"
os.makedirs(out, exist_ok=True)
flags = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -fno-hip-fp32-correctly-rounded-divide-sqrt -fapprox-func -fno-slp-vectorize -gline-tables-only".split()
subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, f"-DLR_VARIANT={mask}", "-c", "-o", f"{out}/g{mask}.o", f"{root}/luisarender_amd/csrc/hip/megapath_variant.hip"])
subprocess.check_call([f"{llvm}/llvm-objcopy", f"--dump-section=.hip_fatbin={out}/g{mask}.bin", f"{out}/g{mask}.o"])
subprocess.check_call([f"{llvm}/clang-offload-bundler", "--type=o", f"--input={out}/g{mask}.bin", "--unbundle", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={out}/g{mask}.co"])
dis = subprocess.run([f"{llvm}/llvm-objdump", "-d", "-l", f"{out}/g{mask}.co"], capture_output=True, text=True).stdout
by_line, by_file, where = collections.Counter(), collections.Counter(), (None, None)
for line in dis.splitlines():
    if line.startswith("; "):
        m = re.match(r"; (\S+):(\d+)", line)
        if m:
            where = (os.path.basename(m.group(1)), int(m.group(2)))
        continue
    if line.strip().startswith("v_"):
        by_line[where] += 1
        by_file[where[0]] += 1
print("VALU instructions by file:", dict(by_file.most_common()))
for (f, n), c in by_line.most_common(top):
    src = open(f"{root}/luisarender_amd/csrc/hip/{f}").read().splitlines()[n - 1].strip()[:110] if f and os.path.exists(f"{root}/luisarender_amd/csrc/hip/{f}") else ""
    print(f"{c:6d}  {f}:{n:<5d} {src}")
