mkdir -p gpurun_out
( time timeout 900 python -m pytest tests -m gpu -x -q --durations=8 ) > gpurun_out/gpu_tests.log 2>&1
tail -15 gpurun_out/gpu_tests.log
ab() { python bench.py --workload $1 --spp $2 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$3', '$1', round(d['value'], 1), 'Msamples/s', round(d['ms_per_step'], 1), 'ms/step')"; }
{
LR_BVH_SWEEP=0 LR_BVH_COLLAPSE=0 ab c2 256 old
ab c2 256 new
LR_BVH_DFS=1 ab c2 256 new_dfs
LR_BVH_SWEEP=0 LR_BVH_COLLAPSE=1 ab c2 256 binned_collapse
LR_BVH_COLLAPSE=0 ab c2 256 sweep_greedy
LR_BVH_SWEEP=0 LR_BVH_COLLAPSE=0 ab c5 256 old
ab c5 256 new
} > gpurun_out/ab_bvh.log 2>&1
cat gpurun_out/ab_bvh.log
