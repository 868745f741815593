#!/bin/bash
# round 3, call H: GPU suite at the current source, the default bench line, C5 as the bench workload (2048 spp, wavefront mode),
# rocprofv3 kernel trace + PMC passes of C2 at 1024 spp (raw .db files stay under gpurun_out/), wavefront timeline of C5 at 512 spp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03h; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -s > gpurun_out/r03h/gpu_tests.log 2>&1; grep -E "passed|failed|C2-class|FAILED|^E  " gpurun_out/r03h/gpu_tests.log | tail -10
( time timeout 900 python bench.py > gpurun_out/r03h/bench.json 2> gpurun_out/r03h/bench.err ) 2>&1 | grep real
timeout 600 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-extra > gpurun_out/r03h/bench_c5.json 2> gpurun_out/r03h/bench_c5.err
python - <<'PY'
import json
for f in ("bench", "bench_c5"):
    d = json.loads([l for l in open(f"gpurun_out/r03h/{f}.json") if l.startswith("{")][-1])
    print(f, round(d["value"], 1), "frac", d["roofline"].get("frac"), "valu", (d["roofline"].get("valu") or {}).get("wave_instr_per_sample"), "parity", {k: v for k, v in (d.get("parity") or {}).items() if k in ("rel_l1", "rmse_over_mean", "flip")})
    for e in d.get("extra_configs", []):
        print("   ", e["workload"][:34], e["sampler"], e["spp_timed"], round(e["value"], 1))
PY
tools/profile_c2.sh r03h 1024 > gpurun_out/r03h/profile.log 2>&1
python tools/summarize_prof.py gpurun_out/prof_r03h gpurun_out/r03h/c2_1024spp.json 1073741824 > /dev/null 2>&1; head -c 1500 gpurun_out/r03h/c2_1024spp.json
( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/r03h/trace_c5 -o trace -- python $R/tools/c5_ablation.py 512 full > $R/gpurun_out/r03h/trace_c5.log 2>&1 )
python tools/wf_trace.py gpurun_out/r03h/trace_c5 | head -14 | tee gpurun_out/r03h/wf_trace_c5.txt
