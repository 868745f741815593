#!/usr/bin/env python3
"""scratch (spill / stack) instructions per function of a disassembled code object, and inside the traversal loop of the kernel
(between the first global_load_lds and the loop's backward branches):  tools/isa_scratch.py <file.s>"""
import re
import sys
fn = None
counts = {}
lines = open(sys.argv[1]).read().split("\n")
for l in lines:
    m = re.match(r"^[0-9a-f]+ <(.+)>:", l)
    if m:
        fn = m.group(1)[:70]
        counts.setdefault(fn, [0, 0, 0])
        continue
    if fn is None:
        continue
    t = l.strip()
    if not t or t.startswith("//"):
        continue
    counts[fn][2] += 1
    if t.startswith("scratch_load"):
        counts[fn][0] += 1
    elif t.startswith("scratch_store"):
        counts[fn][1] += 1
for f, (ld, st, n) in counts.items():
    print(f"{f:70s} instructions {n:6d}  scratch loads {ld:4d} stores {st:4d}")
