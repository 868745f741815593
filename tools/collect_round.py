#!/usr/bin/env python3
"""Copies what tools/gpu_round_check.sh <tag> left under gpurun_out/<tag>/ to profiles/<tag>_* under the names profiles/README.md lists, cuts the
line's per-configuration roofline blocks into files of their own, and prints the round's numbers table.     python tools/collect_round.py r06_final"""
import json
import os
import shutil
import sys

tag = sys.argv[1]
src, dst = os.path.join("gpurun_out", tag), "profiles"


def last_line(path):
    return json.loads([l for l in open(path) if l.startswith("{")][-1])


line = last_line(os.path.join(src, "bench_c2.json"))
json.dump(line, open(os.path.join(dst, f"{tag}_bench_c2_1gpu.json"), "w"), indent=1)
json.dump(last_line(os.path.join(src, "bench_c5.json")), open(os.path.join(dst, f"{tag}_bench_c5_1gpu.json"), "w"), indent=1)
json.dump(last_line(os.path.join(src, "bench_forced_collective.json")), open(os.path.join(dst, f"{tag}_bench_forced_collective_1rank.json"), "w"), indent=1)
for a, b in (("c2_1024spp.json", f"{tag}_c2_1024spp.json"), ("wf_trace_c5.txt", f"{tag}_wf_trace_c5_2048spp.txt"), ("shard_probe.txt", f"{tag}_shard_probe_c2.txt")):
    shutil.copy(os.path.join(src, a), os.path.join(dst, b))
shutil.copy(os.path.join(src, "c2_1024spp.json"), os.path.join(dst, "pmc_c2.json"))
with open(os.path.join(dst, f"{tag}_schedulers.txt"), "w") as f:
    for name in ("ab_sched_c2_1024spp.txt", "ab_sched_others_64spp.txt", "stats_lane_c2.txt", "stats_pool_c2.txt"):
        f.write(f"# {name}\n" + open(os.path.join(src, name)).read())
with open(os.path.join(dst, f"{tag}_gpu_tests.txt"), "w") as f:
    f.write("".join(l for l in open(os.path.join(src, "gpu_tests.log")) if "amdgpu.ids" not in l)[-20000:])
names = {"Cornell Box": "c1", "Bedroom-class": "c3", "Camera-class": "c4", "Kitchen-class": "c5", "Contemporary Bathroom-class": "c2_padded_sobol"}
rows = [("C2", line["config"]["workload"], "Independent", line["value"], line["roofline"], line.get("parity"), line.get("cpu_baseline"))]
for e in line["extra_configs"]:
    key = names[e["workload"].split(",")[0].split(" (")[0]]
    json.dump({"workload": e["workload"], "sampler": e["sampler"], "spp_timed": e["spp_timed"], "value_msamples_per_s": e["value"], "source_hash": line.get("source_hash"), "roofline": e["roofline"]},
              open(os.path.join(dst, f"{tag}_{key}_roofline.json"), "w"), indent=1)
    rows.append((key, e["workload"], e["sampler"], e["value"], e["roofline"], e.get("parity"), e.get("cpu_baseline")))
print("source_hash", line.get("source_hash"), " ms_per_step", line["ms_per_step"], " cpu_reference", {k: line["cpu_reference"][k] for k in ("value", "cores", "kind")})
for key, wl, sampler, value, r, parity, cpu in rows:
    w, v = r.get("waves", {}), r.get("valu", {})
    print(f"{key:16s} {sampler:12s} {value:8.1f} Msamples/s  kernel {r['kernel'][:44]:44s} frac {r['frac']:.3f}  issue {v.get('issue_frac', 0):.2f}  "
          f"wait/stall/fly {w.get('waiting_at_waitcnt', 0):.2f}/{w.get('issue_stalled', 0):.2f}/{w.get('instruction_in_flight', 0):.2f}  lanes {r['lanes']['trace']:.2f}/{r['lanes']['shade']:.2f}  "
          f"parity rel-L1 {parity['rel_l1']:.2e} flip {parity.get('flip', float('nan')):.2e}  cpu {cpu['value']:.2f} ({cpu['cores']} threads)")
m = last_line(os.path.join(src, "bench_forced_collective.json"))
print("forced collective:", m["value"], {k: m["multi_gpu"][k] for k in ("reduce_ms", "reduce_bytes", "reduced_film_equals_1gpu_render") if k in m["multi_gpu"]})
for e in m.get("extra_configs", []):
    print("   ", e["workload"][:20], e["value"], {k: e.get("multi_gpu", {}).get(k) for k in ("reduce_ms", "reduce_bytes", "reduced_film_equals_1gpu_render")})
