#!/usr/bin/env python3
"""Loop structure of a disassembled megakernel variant (llvm-objdump -d of the .co): for every backward branch, the extent of
the loop and how many spill (scratch) / LDS-direct-load / VALU instructions its body holds.  tools/kernel_resources.sh leaves the
code objects in /tmp/kres:   llvm-objdump -d /tmp/kres/v0.co > /tmp/kres/v0.s && python tools/isa_loops.py /tmp/kres/v0.s"""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
addr, base = {}, None
for i, l in enumerate(lines):
    m = re.match(r'([0-9a-f]{16}) <(.*)>:', l)
    if m and ('megapath_kernel' in m.group(2) or 'megapool_kernel' in m.group(2) or 'megavpt_kernel' in m.group(2)):
        base = int(m.group(1), 16)
    m = re.search(r'//\s+([0-9A-F]{12}):', l)
    if m:
        addr[int(m.group(1), 16)] = i
VALU = re.compile(r'\s+v_')
for i, l in enumerate(lines):
    m = re.search(r'(s_cbranch\w+|s_branch)\s+\S+\s+//\s+([0-9A-F]{12}):\s+\S+\s+<[^+]+\+0x([0-9A-Fa-f]+)>', l)
    if not m or base is None:
        continue
    cur, tgt = int(m.group(2), 16), base + int(m.group(3), 16)
    if tgt >= cur or tgt not in addr:
        continue
    body = lines[addr[tgt]:i + 1]
    print("loop lines %6d-%6d  instr %5d  scratch st %3d ld %3d  global_load_lds %d  valu %5d  waitcnt %3d" % (
        addr[tgt], i, len(body), sum('scratch_store' in b for b in body), sum('scratch_load' in b for b in body),
        sum('global_load_lds' in b for b in body), sum(bool(VALU.match(b)) for b in body), sum('s_waitcnt' in b for b in body)))
