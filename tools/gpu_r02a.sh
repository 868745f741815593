set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(time timeout 1700 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -60) > gpurun_out/r02a_gputests.log 2>&1
python bench.py --steps 5 --warmup 2 > gpurun_out/r02a_bench.json 2> gpurun_out/r02a_bench.err
tail -3 gpurun_out/r02a_gputests.log; cat gpurun_out/r02a_bench.json
