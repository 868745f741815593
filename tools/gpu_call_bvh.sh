# A/B of host BVH builder knobs on the GPU box (accel.cpp reads LR_BVH_* from the environment)
mkdir -p gpurun_out
ab() { python bench.py --workload $1 --spp $2 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$3', '$1', round(d['value'], 1), 'Msamples/s', round(d['ms_per_step'], 1), 'ms/step')"; }
{
LR_BVH_REINSERT=0 ab c2 256 reinsert0
LR_BVH_REINSERT=1 ab c2 256 reinsert1
LR_BVH_REINSERT=2 ab c2 256 reinsert2
LR_BVH_REINSERT=1 LR_BVH_DFS=1 ab c2 256 reinsert1_dfs
LR_BVH_REINSERT=0 ab c5 256 reinsert0
LR_BVH_REINSERT=1 ab c5 256 reinsert1
LR_BVH_REINSERT=0 ab c3 256 reinsert0
LR_BVH_REINSERT=1 ab c3 256 reinsert1
} > gpurun_out/ab_bvh2.log 2>&1
cat gpurun_out/ab_bvh2.log
