cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "counting_twins or variant_selection or alpha" 2>&1 | tail -4
python tools/c5_ablation.py 64 alpha_only basic_no_alpha 2>&1 | tail -3
