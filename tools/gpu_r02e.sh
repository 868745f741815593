cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_ref_golden.py -m gpu -q -k "alpha or kitchen or cornell_same or volumetric or reference_frame or counting_twins" 2>&1 | tail -4
python tools/c5_ablation.py 64 full no_layered alpha_only basic_no_alpha 2>&1 | tail -5
bash tools/ab.sh 256 base 2>&1 | tail -2
