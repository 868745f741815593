#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python bench.py > gpurun_out/r02at_bench.json 2> gpurun_out/r02at_bench.err; tail -c 300 gpurun_out/r02at_bench.json
