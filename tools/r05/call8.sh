#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05g; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -s -k "converges or shipped_pool or multi_gpu_path" > $O/gpu_tests_new.log 2>&1; grep -E "passed|failed|FAILED|^E  |Fatal|Error|c5 192|shipped vs" $O/gpu_tests_new.log | tail -30
( time timeout 900 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err ) 2>&1 | grep real
python - <<PY
import json
d = json.loads([l for l in open("$O/bench_c2.json") if l.startswith("{")][-1])
r = d["roofline"]
print("c2", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 1), "frac", r.get("frac"), "valu", {k: v for k, v in (r.get("valu") or {}).items() if k in ("issue_frac", "pmc_busy", "wave_instr_per_sample", "pmc_wait_any_over_wave_cycles")}, "lanes", {k: v for k, v in (r.get("lanes") or {}).items() if k != "note"},
      "l2", (r.get("l2") or {}).get("frac"), "parity", {k: v for k, v in (d.get("parity") or {}).items() if k in ("rel_l1", "rmse_over_mean", "flip")})
for e in d.get("extra_configs", []):
    print("   ", e["workload"][:34], e["sampler"], e["spp_timed"], "of", e.get("spp_config"), round(e["value"], 1), {k: v for k, v in (e.get("parity") or {}).items() if k in ("rel_l1", "flip")})
PY
