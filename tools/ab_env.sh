#!/bin/bash
# A/B of an environment variable on the bench: tools/ab_env.sh <spp> <VAR> <value>...
SPP=$1; VAR=$2; shift; shift
for v in "$@"; do
  export $VAR=$v
  python bench.py --spp $SPP --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$VAR=$v', round(d['value'], 1), 'Msamples/s', round(d['ms_per_step'], 1), 'ms/step')"
done
