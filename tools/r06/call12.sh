#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06n; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "lamp_lit_fog" 2>&1 | grep  'vpt lamp-lit\|AssertionError\|passed\|failed' > $O/fog.txt
