#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "reduce or shard"; echo "exit code $?"
timeout 600 python -m pytest tests/test_ref_golden.py tests/test_gpu_parity.py -m gpu -x -q -k "nested"; echo "exit code $?"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()"; echo "exit code $?"
} > gpurun_out/r02n.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname" gpurun_out/r02n.txt | tail -20
