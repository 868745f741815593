#!/bin/bash
# round 3, call C: full GPU suite with the new parity tests (full-size C1 vs the reference's frame, IEEE build on C2, C-ABI tree
# validation, CLI forced collective), the default bench line (parity / rays_per_s / extra configs), the collective path with one rank
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03c
timeout 900 python -m pytest tests -m gpu -q -s > gpurun_out/r03c/gpu_tests.log 2>&1; grep -E "passed|failed|C1 full|C2 window|FAILED|Error" gpurun_out/r03c/gpu_tests.log | tail -12
( time timeout 900 python bench.py > gpurun_out/r03c/bench.json 2> gpurun_out/r03c/bench.err ) 2>&1 | grep real
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r03c/bench.json") if l.startswith("{")][-1])
print("value", d["value"], "parity", d.get("parity"), "rays_per_s", d.get("rays_per_s"), "mpl", d.get("mean_path_length"))
for e in d.get("extra_configs", []):
    print(e["workload"][:40], e["sampler"], e["spp_timed"], round(e["value"], 1), e.get("parity"))
PY
LR_BENCH_FORCE_COLLECTIVE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --workload c1 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-extra > gpurun_out/r03c/bench_force.json 2> gpurun_out/r03c/bench_force.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r03c/bench_force.json") if l.startswith("{")][-1])
print("forced collective:", d["value"], d["config"]["collective"], d.get("multi_gpu"))
PY
hipcc --offload-arch=gfx950 -O3 tools/valu_peak2.hip -o gpurun_out/r03c/valu_peak2 2>/dev/null && timeout 300 gpurun_out/r03c/valu_peak2 > gpurun_out/r03c/valu_peak.json; rm -f gpurun_out/r03c/valu_peak2
grep -E "cndmask|sdwa|v_or|v_and" gpurun_out/r03c/valu_peak.json
