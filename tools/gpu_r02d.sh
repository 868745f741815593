cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(time python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "full_size_c3 or counting_twins or diagonals or bound_film or c_abi_single" 2>&1 | grep -E "c3:|c4:|c5:|passed|failed|Error|assert|FAILED" | head -40) 2>&1
