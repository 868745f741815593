#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=gpurun_out/r05h; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -s -k "converges or shipped_pool or multi_gpu_path or kitchen_class or full_size" > $O/gpu_tests_new.log 2>&1; grep -E "passed|failed|FAILED|^E  |Fatal|Error|c5 |shipped vs" $O/gpu_tests_new.log | tail -30
timeout 600 python tools/c5_ablation.py 512 > $O/c5_ablation_512.txt 2>&1; grep -v amdgpu $O/c5_ablation_512.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $R/$O/trace_c5 -o trace -- python $R/tools/c5_ablation.py 2048 full > $R/$O/trace_c5.log 2>&1 )
python tools/wf_trace.py $O/trace_c5 | head -20 | tee $O/wf_trace_c5_2048.txt
find $O/trace_c5 -name "*.db" -size +20M -delete 2>/dev/null
