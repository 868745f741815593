#!/bin/bash
# round 3, call N: GPU suite (TileShared, per-kind heavy kernels, stack-fast), C5 with slices of a quarter of the free HBM, C2 / C5 bench lines
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03n
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r03n/gpu_tests.log 2>&1; grep -E "passed|failed|FAILED|^E  " gpurun_out/r03n/gpu_tests.log | tail -6
{ timeout 300 python tools/c5_ablation.py 512 full; WF_SLICE_PATHS=134217728 timeout 300 python tools/c5_ablation.py 512 full; } 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03n/c5.txt
timeout 600 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-extra > gpurun_out/r03n/bench_c5.json 2> gpurun_out/r03n/bench_c5.err
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-extra > gpurun_out/r03n/bench_c2.json 2> gpurun_out/r03n/bench_c2.err
python - <<'PY'
import json
for f in ("bench_c5", "bench_c2"):
    d = json.loads([l for l in open(f"gpurun_out/r03n/{f}.json") if l.startswith("{")][-1])
    print(f, round(d["value"], 1), d["ms_per_step"])
PY
rocm-smi --showmeminfo vram 2>/dev/null | head -8
