#!/usr/bin/env python3
"""Timeline of a wavefront-mode render from a rocprofv3 --kernel-trace database: per kernel launches / total / mean duration, and the
idle gaps between consecutive kernels of the stream (what a round of { heavy kernel -> continuation pass } costs beyond its work).
    rocprofv3 --kernel-trace -d gpurun_out/x -o trace -- python tools/c5_ablation.py 64 full ; python tools/wf_trace.py gpurun_out/x"""
import os
import sqlite3
import sys

dbs = [os.path.join(r, f) for r, _, fs in os.walk(sys.argv[1]) for f in fs if f.endswith(".db")]
db = sqlite3.connect(dbs[0])
rows = list(db.execute("select name, start, end, scratch_size from kernels order by start"))
agg = {}
for name, s, e, scratch in rows:
    short = name.split("(")[0][-60:]
    a = agg.setdefault(short, [0, 0.0, scratch])
    a[0] += 1
    a[1] += (e - s) / 1e6
print(f"{len(rows)} kernel launches, span {(rows[-1][2] - rows[0][1]) / 1e6:.1f} ms, busy {sum(a[1] for a in agg.values()):.1f} ms")
for k, (n, ms, scratch) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:62s} x{n:5d}  total {ms:9.2f} ms  mean {ms / n * 1e3:9.1f} us  scratch/lane {scratch}")
gaps = [(rows[i + 1][1] - rows[i][2]) / 1e3 for i in range(len(rows) - 1)]
# a launch that starts more than 5 us before its predecessor in start order has ended runs on another stream (a memset of the host
# side under a long kernel): not a gap of this stream
overlapped = sum(1 for g in gaps if g < -5.0)
gaps = [max(g, 0.0) if g >= -5.0 else 0.0 for g in gaps]
gaps_small = [g for g in gaps if g < 5000]
print(f"gaps between consecutive kernels: mean {sum(gaps_small) / max(len(gaps_small), 1):.1f} us, total {sum(gaps_small) / 1e3:.1f} ms (gaps above 5 ms -- host work -- left out: {len(gaps) - len(gaps_small)}; launches under a running kernel of another stream: {overlapped})")
by_prev = {}
for i, g in enumerate(gaps):
    if g < 5000:
        k = rows[i][0].split("(")[0][-40:] + " -> " + rows[i + 1][0].split("(")[0][-40:]
        a = by_prev.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += g
for k, (n, us) in sorted(by_prev.items(), key=lambda kv: -kv[1][1])[:8]:
    print(f"  {k:90s} x{n:5d} mean {us / n:8.1f} us")
# per ROUND of a slice (the launches between two camera passes, in order): mean duration of the continuation pass and of the heavy kernels --
# how much of the frame is the tail of near-empty rounds
rounds = {}
index = {}
for name, s, e, _ in rows:
    short = name.split("(")[0]
    if "kernel<" not in short:
        continue
    is_cont = "megap" in short and any(f"<{m}u>" in short for m in range(2048, 4096)) or any(f"<{m}u>" in short for m in range(6144, 8192))
    is_camera = ("megapool_kernel" in short or "megapath_kernel" in short) and not is_cont
    if is_camera:
        index = {}
        continue
    key = "continuation" if is_cont else short.split("lrd::")[-1]
    i = index.get(key, 0)
    index[key] = i + 1
    a = rounds.setdefault((key, i), [0, 0.0])
    a[0] += 1
    a[1] += (e - s) / 1e6
if rounds:
    keys = sorted({k for k, _ in rounds})
    print("per round of a slice, mean ms:  " + "  ".join(f"{k[-22:]:>22s}" for k in keys))
    for i in range(max(i for _, i in rounds) + 1):
        print(f"  round {i:2d}                      " + "  ".join(f"{rounds[k, i][1] / rounds[k, i][0]:22.3f}" if (k, i) in rounds else " " * 22 for k in keys))
