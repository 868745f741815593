#!/usr/bin/env python3
"""Runs of consecutive v_cndmask_b32_e32 (implicit VCC) in a disassembled kernel: on gfx950 such an instruction issued right behind
another one costs ~19 cycles instead of 2-4 (tools/valu_peak2.hip, profiles/archive/r04h_cndmask_forms.json); other VALU work between them,
or the VOP3 form, does not.    llvm-objdump -d x.co > x.s && python tools/isa_cndmask_runs.py x.s [context lines]"""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ins = []
for i, l in enumerate(lines):
    m = re.match(r'\s+([a-z_0-9]+)\s*(.*?)\s*//', l)
    if m:
        ins.append((i + 1, m.group(1), m.group(2)))
runs, cur = [], []
for ln, op, args in ins:
    if op == 'v_cndmask_b32_e32':
        cur.append(ln)
        continue
    if op.startswith('v_'):  # another VALU instruction ends the run; SALU / waitcnt / memory instructions do not
        if len(cur) >= 2:
            runs.append(cur)
        cur = []
print(f"{len(runs)} runs, {sum(len(r) - 1 for r in runs)} instructions behind another one: " + ", ".join(f"{r[0]}x{len(r)}" for r in runs))
if ctx:
    for r in runs:
        print("----")
        for l in lines[r[0] - 1 - ctx:r[-1] + 1]:
            print(l.split('//')[0].rstrip()[:100])
