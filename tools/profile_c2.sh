#!/bin/bash
# Profiling recipe for the bench workload (run on the GPU box through gpurun):
#   [WL=c2|c3|c4|c5] [SAMPLER=PaddedSobol|Sobol] tools/profile_c2.sh <tag> <spp>
#   1. rocprofv3 --kernel-trace --stats    -> per-kernel durations (must agree with bench.py's HIP-event time)
#   2. separate --pmc passes (FETCH_SIZE / WRITE_SIZE do not fit in one pass; SQ + TCC hit counters in a third)
# Summaries are written under gpurun_out/prof_<tag>/ and the interesting files copied to profiles/ by hand.
set -u
TAG=${1:-r01}
SPP=${2:-64}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --workload ${WL:-c2} --spp $SPP --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-extra --no-stats ${SAMPLER:+--sampler $SAMPLER}"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o pmc -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o pmc -- $CMD > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/pmc_sq -o pmc -- $CMD > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_RD --kernel-trace -d $OUT/pmc_tcc -o pmc -- $CMD > $OUT/pmc_tcc.log 2>&1
find $OUT -name "*.csv" | head -40
tail -2 $OUT/trace.log
