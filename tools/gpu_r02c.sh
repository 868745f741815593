cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(time python bench.py --steps 3 --warmup 1 > gpurun_out/r02c_bench.json 2> gpurun_out/r02c_bench.err) 2>&1 | tail -3
python - <<'PY'
import json
d = json.load(open('gpurun_out/r02c_bench.json'))
print(d['value'], d['ms_per_step'])
print(json.dumps(d['roofline'], indent=1))
print(json.dumps(d.get('extra_configs'), indent=1)[:3000])
PY
tail -5 gpurun_out/r02c_bench.err
