#!/bin/bash
# One gpurun call that produces everything a round's evidence needs, at ONE source hash (run on the GPU box):
#   gpurun --timeout 3000 -- 'bash tools/gpu_round_check.sh <tag>'
#   1. the GPU suite + smoke                                 -> gpurun_out/<tag>/gpu_tests.log
#   2. the default bench line (what the driver runs)          -> gpurun_out/<tag>/bench_c2.json  (copy to profiles/<tag>_bench_c2_1gpu.json)
#   3. C5 as the bench workload (2048 spp, wavefront mode)    -> gpurun_out/<tag>/bench_c5.json
#   4. rocprofv3 kernel trace + PMC passes of C2 at 1024 spp  -> gpurun_out/prof_<tag>/ (raw .db files stay there), summary c2_1024spp.json
#   5. wavefront timeline of C5 at 512 spp                    -> gpurun_out/<tag>/wf_trace_c5.txt
#   6. shard probe of C2 (1 GPU standing in for rank 0 of N)  -> gpurun_out/<tag>/shard_probe.txt
#   7. (round 4) the two schedulers side by side at the bench size, the forced one-rank collective through torch.distributed.run
TAG=${1:-rXX}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=gpurun_out/$TAG
timeout 1500 python -m pytest tests -m gpu -q -s > $O/gpu_tests.log 2>&1; grep -E "passed|failed|FAILED|^E  |Fatal" $O/gpu_tests.log | tail -6
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
( time timeout 900 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err ) 2>&1 | grep real
timeout 600 python bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-extra --no-stats > $O/bench_c5.json 2> $O/bench_c5.err
python - <<PY
import json
for f in ("bench_c2", "bench_c5"):
    d = json.loads([l for l in open("$O/%s.json" % f) if l.startswith("{")][-1])
    r = d["roofline"]
    print(f, round(d["value"], 1), "ms/step", round(d["ms_per_step"], 1), "frac", r.get("frac"), "valu", (r.get("valu") or {}).get("issue_frac"), "lanes", {k: v for k, v in (r.get("lanes") or {}).items() if k != "note"},
          "l2", (r.get("l2") or {}).get("frac"), "parity", {k: v for k, v in (d.get("parity") or {}).items() if k in ("rel_l1", "rmse_over_mean", "flip")}, "hash", d.get("source_hash"))
    for e in d.get("extra_configs", []):
        print("   ", e["workload"][:34], e["sampler"], e["spp_timed"], "of", e.get("spp_config"), round(e["value"], 1), {k: v for k, v in (e.get("parity") or {}).items() if k in ("rel_l1", "flip")})
PY
tools/profile_c2.sh $TAG 1024 > $O/profile.log 2>&1
python tools/summarize_prof.py gpurun_out/prof_$TAG $O/c2_1024spp.json 1073741824 > /dev/null 2>&1; python -c "
import json; d = json.load(open('$O/c2_1024spp.json')); print({k: d[k] for k in d if k not in ('top_kernels', 'kernel_ms_all', 'counters_mean_per_launch')})"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/$O/trace_c5 -o trace -- python $R/tools/c5_ablation.py 2048 full > $R/$O/trace_c5.log 2>&1 )
python tools/wf_trace.py $O/trace_c5 | head -40 | tee $O/wf_trace_c5.txt
timeout 600 python tools/shard_probe.py 1024 2>&1 | grep -v amdgpu.ids | tee $O/shard_probe.txt
timeout 600 python tools/ab_sched.py 1024 c2 2>&1 | grep "^c2" | tee $O/ab_sched_c2_1024spp.txt
timeout 600 python tools/ab_sched.py 64 c1 c3 c4 c5 2>&1 | grep "^c[0-9]" | tee $O/ab_sched_others_64spp.txt
LRHIP_SCHEDULER=legacy timeout 300 python tools/gpu_stats.py 1024 c2 2>&1 | grep -v amdgpu | tail -8 > $O/stats_lane_c2.txt; LRHIP_SCHEDULER=pool timeout 300 python tools/gpu_stats.py 1024 c2 2>&1 | grep -v amdgpu | tail -8 > $O/stats_pool_c2.txt
grep -h "utilisation" $O/stats_lane_c2.txt $O/stats_pool_c2.txt
LR_BENCH_FORCE_COLLECTIVE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-extra --no-stats > $O/bench_forced_collective.json 2> $O/bench_forced_collective.err
python -c "
import json
d = json.loads([l for l in open('$O/bench_forced_collective.json') if l.startswith('{')][-1]); print('forced 1-rank collective:', round(d['value'], 1), d['config'].get('collective'), d.get('multi_gpu'))"
find gpurun_out/prof_$TAG $O/trace_c5 -name "*.db" -size +20M -delete 2>/dev/null
