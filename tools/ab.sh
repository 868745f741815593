#!/bin/bash
# A/B of kernel build variants on the GPU box: [WL=c2] tools/ab.sh <spp> <variant>...   (variant "base" = shipped lib)
SPP=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset LRHIP_LIB; else export LRHIP_LIB=$PWD/luisarender_amd/lib/variants/liblrhip_$v.so; fi
  python bench.py --workload ${WL:-c2} --spp $SPP --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-extra --no-stats ${SAMPLER:+--sampler $SAMPLER} 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', round(d['value'], 1), 'Msamples/s', round(d['ms_per_step'], 1), 'ms/step')"
done
