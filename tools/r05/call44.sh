#!/bin/bash
# the driver's command a second time on another box at the final hash: how far two boxes are apart
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05zj; export TMPDIR=/tmp
( time timeout 600 python bench.py > gpurun_out/r05zj/bench_c2_run2.json 2> gpurun_out/r05zj/bench_c2_run2.err ) 2>&1 | grep real
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r05zj/bench_c2_run2.json") if l.startswith("{")][-1])
print(round(d["value"], 1), d["source_hash"], [(e["workload"][:12], e["sampler"][:6], round(e["value"], 1)) for e in d["extra_configs"]])
PY
