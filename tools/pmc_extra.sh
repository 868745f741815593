#!/bin/bash
# extra PMC passes (instruction cache, LDS / SALU / VMEM issue, flat / write instructions, LDS conflicts) on a bench workload:
#   [LRHIP_SCHEDULER=legacy] tools/pmc_extra.sh <tag> <spp> [workload]
set -u
TAG=${1:-x}; SPP=${2:-256}; WL=${3:-c2}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmcx_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --workload $WL --spp $SPP --steps 1 --warmup 1 --no-cpu-baseline --no-pmc --no-extra --no-stats"
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ --kernel-trace -d $OUT/ic -o pmc -- $CMD > $OUT/ic.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES --kernel-trace -d $OUT/iss -o pmc -- $CMD > $OUT/iss.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_FLAT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --kernel-trace -d $OUT/mem -o pmc -- $CMD > $OUT/mem.log 2>&1
timeout 300 rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace -d $OUT/if -o pmc -- $CMD > $OUT/if.log 2>&1
python - << PY
import sqlite3, glob, json
out = {}
for sub in ("ic", "iss", "mem", "if"):
    for db in glob.glob("$OUT/%s/*results.db" % sub) + glob.glob("$OUT/%s/*/*results.db" % sub):
        d = sqlite3.connect(db)
        try:
            for name, value, n in d.execute("select counter_name, avg(value), count(*) from counters_collection where kernel_name like '%megap%_kernel%' group by counter_name"):
                out[name] = value
            for name, ms in d.execute("select name, avg(end-start)/1e6 from kernels where name like '%megap%_kernel%' group by name"):
                out["kernel_ms"] = ms
        except Exception as e:
            out[sub + "_error"] = str(e)
print(json.dumps(out, indent=1))
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
PY
find $OUT -name "*.db" -delete
