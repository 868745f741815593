#!/bin/bash
# the bench's collective path with one rank under torchrun (the launcher line the driver uses), + the new fixtures
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
LR_BENCH_FORCE_COLLECTIVE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 2 --warmup 1 --spp 64 --no-cpu-baseline --no-pmc --no-extra 2>&1 | tail -3
timeout 600 python bench.py --gpus 1 --steps 2 --warmup 1 --spp 64 --no-cpu-baseline --no-pmc --no-extra 2>&1 | tail -1 | cut -c1-400
timeout 900 python -m pytest tests/test_ref_golden.py tests/test_gpu_parity.py -m gpu -x -q -k "nested or reduce or shard" 2>&1 | tail -3
} > gpurun_out/r02m.txt 2>&1
cat gpurun_out/r02m.txt
