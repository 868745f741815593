#!/bin/bash
# PMC passes of C2 (pool) for several builds: tools/r05/pmc_ab.sh <tag> <spp> <build>...   -> gpurun_out/<tag>/pmc_<build>.json
TAG=$1; SPP=$2; shift; shift
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for b in "$@"; do
  if [ "$b" = base ]; then unset LRHIP_LIB; else export LRHIP_LIB=$R/luisarender_amd/lib/variants/liblrhip_$b.so; fi
  CMD="python $R/bench.py --workload ${WL:-c2} --spp $SPP --steps 1 --warmup 0 --no-cpu-baseline --no-pmc --no-extra --no-stats"
  P=/tmp/pmc_$b; rm -rf $P
  ( cd /tmp
    timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $P/sq -o pmc -- $CMD > $P.sq.log 2>&1
    timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace -d $P/tcc -o pmc -- $CMD > $P.tcc.log 2>&1
    timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P/fetch -o pmc -- $CMD > $P.fetch.log 2>&1
    timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $P/write -o pmc -- $CMD > $P.write.log 2>&1
    timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_IFETCH --kernel-trace -d $P/act -o pmc -- $CMD > $P.act.log 2>&1 )
  python - <<PY
import sqlite3, glob, json
out = {"build": "$b", "spp": $SPP}
for sub in ("sq", "tcc", "fetch", "write", "act"):
    for db in glob.glob("$P/%s/**/*.db" % sub, recursive=True):
        d = sqlite3.connect(db)
        try:
            for name, value in d.execute("select counter_name, avg(value) from counters_collection where kernel_name like '%mega%_kernel%' group by counter_name"):
                out[name] = value
            for name, ms in d.execute("select name, avg(end-start)/1e6 from kernels where name like '%mega%_kernel%' group by name"):
                out["kernel"] = name; out["kernel_ms_" + sub] = ms
        except Exception as e:
            out[sub + "_error"] = str(e)
json.dump(out, open("$O/pmc_$b.json", "w"), indent=1)
wc = out.get("SQ_WAVE_CYCLES", 0) or 1
print("$b", {k: (round(v / wc, 4) if k.startswith(("SQ_ACTIVE", "SQ_WAIT", "SQ_INST_CYCLES")) else v) for k, v in out.items() if k not in ("build", "kernel")})
PY
done
