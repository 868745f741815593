#!/usr/bin/env python3
"""Aggregates rocprofv3 PC-sampling output (GPU box) into a histogram small enough to come home through gpurun_out/.

    python tools/pc_aggregate.py <rocprofv3 output dir> <out.json>

CSV (streamed): one row per sample -> counts by (instruction text, source comment, issued?, instruction type, stall reason).
JSON (streamed with a regex, the file is too large to parse whole): counts by code-object offset where the records carry one.
The first rows / bytes of both are kept verbatim under "peek" so that the schema can be read off the result."""
import collections
import csv
import json
import os
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
out = {"files": [], "peek": {}, "csv": {}, "json_offsets": {}}
for root, _, files in os.walk(src):
    for f in files:
        p = os.path.join(root, f)
        out["files"].append([p, os.path.getsize(p)])
        if "pc_sampling" in f and f.endswith(".csv"):
            kind = "stochastic" if "stochastic" in f else "host_trap"
            counts = collections.Counter()
            dispatch = collections.Counter()
            with open(p, newline="") as fh:
                rd = csv.reader(fh)
                header = next(rd, [])
                out["peek"][f] = {"header": header, "rows": []}
                col = {name: i for i, name in enumerate(header)}
                get = lambda row, name: row[col[name]] if name in col and col[name] < len(row) else ""
                n = 0
                for row in rd:
                    if n < 30:
                        out["peek"][f]["rows"].append(row)
                    n += 1
                    key = (get(row, "Instruction"), get(row, "Instruction_Comment"), get(row, "Wave_Issued_Instruction"), get(row, "Instruction_Type"),
                           get(row, "Stall_Reason"))
                    counts[key] += 1
                    dispatch[get(row, "Dispatch_Id")] += 1
            out["csv"][kind] = {"samples": n, "by_dispatch": dict(dispatch.most_common(20)),
                                "rows": [[*k, v] for k, v in counts.most_common(20000)]}
        if f.endswith(".json"):
            size = os.path.getsize(p)
            offs = collections.Counter()
            peeked = False
            tail = ""
            with open(p, "r", errors="replace") as fh:
                while True:
                    chunk = fh.read(1 << 24)
                    if not chunk:
                        break
                    buf = tail + chunk
                    if not peeked:
                        i = buf.find("pc_sampl")
                        if i >= 0:
                            j = buf.find("code_object_offset", i)
                            out["peek"][f] = {"first_mention": buf[max(0, i - 200):i + 1500], "first_record": buf[max(0, j - 1500):j + 1500] if j >= 0 else ""}
                            peeked = j >= 0
                    for m in re.finditer(r'"code_object_id"\s*:\s*(\d+)\s*,\s*"code_object_offset"\s*:\s*(\d+)', buf):
                        offs[(int(m.group(1)), int(m.group(2)))] += 1
                    tail = buf[-200:]
            # (a match that straddles the 200-byte overlap is counted twice at worst: negligible against millions of samples)
            out["json_offsets"][f] = {"bytes": size, "distinct": len(offs), "rows": [[a, b, n] for (a, b), n in offs.most_common(40000)]}
json.dump(out, open(dst, "w"))
print("aggregated:", {k: v.get("samples") for k, v in out["csv"].items()}, {k: v["distinct"] for k, v in out["json_offsets"].items()})
