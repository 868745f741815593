#!/bin/bash
# Per-instruction stall attribution of a megakernel (GPU box, through gpurun): rocprofv3 PC sampling of one short bench run, aggregated on
# the box into a histogram by code-object offset (the raw sample files are hundreds of MB; the histogram is what comes home).
#   [LIB=<variant name under lib/variants>] [SAMPLER=..] [SCHED=auto|pool|legacy] tools/pc_sample.sh <tag> <workload> <spp> [stochastic|host_trap] [interval]
# stochastic: interval in cycles (a power of two); host_trap: in microseconds.  Output: gpurun_out/<tag>/pcs_<workload>.json (+ .log)
# Offsets are mapped to instructions / sections HERE (tools/pc_report.py) against the disassembly of the same liblrhip.so.
set -u
TAG=$1; WL=$2; SPP=$3; METHOD=${4:-stochastic}; INTERVAL=${5:-1048576}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
if [ -n "${LIB:-}" ]; then export LRHIP_LIB=$REPO/luisarender_amd/lib/variants/liblrhip_$LIB.so; fi
UNIT=cycles; [ "$METHOD" = host_trap ] && UNIT=time
D=/tmp/pcs_${TAG}_$WL; rm -rf $D; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --workload $WL --spp $SPP --steps 1 --warmup 0 --no-cpu-baseline --no-pmc --no-extra --no-stats --scheduler ${SCHED:-auto} ${SAMPLER:+--sampler $SAMPLER}"
timeout 600 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $UNIT --pc-sampling-method $METHOD --pc-sampling-interval $INTERVAL \
    --kernel-trace --output-format csv json -d $D -o pcs -- $CMD > $OUT/pcs_$WL.log 2>&1
echo "rocprofv3 rc=$?" >> $OUT/pcs_$WL.log
find $D -type f -printf "%s %p\n" >> $OUT/pcs_$WL.log
python $REPO/tools/pc_aggregate.py $D $OUT/pcs_$WL.json >> $OUT/pcs_$WL.log 2>&1
tail -5 $OUT/pcs_$WL.log
