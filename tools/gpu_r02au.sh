#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_nested -o t -- python $R/bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline > $R/gpurun_out/r02au.txt 2>&1
grep -o '"traffic_source": "[^"]*"' $R/gpurun_out/r02au.txt; grep -o '"value": [0-9.]*' $R/gpurun_out/r02au.txt | head -1; grep -o '"frac": [0-9.]*' $R/gpurun_out/r02au.txt | head -1
rm -rf $R/gpurun_out/prof_nested
