cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests/test_reference_step_gpu.py tests/test_full_step_gpu.py tests/test_transformer_gpu.py -x -q -m gpu 2>&1 | tail -4
bash tools/ab_step.sh CODA_ATTN_DKV_X3 2 40 0 1
