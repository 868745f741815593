#!/bin/bash
# round-2 wrap-up: full GPU suite, smoke, the bench line, the rocprofv3 profile of the same build
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02as_pytest.txt 2>&1; tail -3 gpurun_out/r02as_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02as_smoke.txt 2>&1; tail -1 gpurun_out/r02as_smoke.txt
timeout 1200 python bench.py > gpurun_out/r02as_bench.json 2> gpurun_out/r02as_bench.err; tail -c 600 gpurun_out/r02as_bench.json
bash tools/profile_c2.sh r02j 1024 > /dev/null 2>&1
python tools/summarize_prof.py gpurun_out/prof_r02j gpurun_out/r02j_c2_1024spp.json 1073741824 2>&1 | tail -2
WL=c5 bash tools/profile_c2.sh r02j_c5 256 > /dev/null 2>&1
python tools/summarize_prof.py gpurun_out/prof_r02j_c5 gpurun_out/r02j_c5_256spp.json 235929600 2>&1 | tail -2
cp gpurun_out/prof_r02j/trace/*stats*.csv gpurun_out/ 2>/dev/null
rm -rf gpurun_out/prof_r02j/*/*.db gpurun_out/prof_r02j_c5/*/*.db 2>/dev/null
ls gpurun_out/prof_r02j/trace | head
for w in c3 c4 c5; do timeout 900 python bench.py --workload $w --no-pmc --no-extra > gpurun_out/r02j_bench_$w.json 2>/dev/null; tail -c 200 gpurun_out/r02j_bench_$w.json; done
