#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03q
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r03q/gpu_tests.log 2>&1; grep -E "passed|failed|FAILED|^E  |Fatal" gpurun_out/r03q/gpu_tests.log | tail -6
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
