cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/ab_step.sh CODA_ATTN_DQ_X3 2 40 0 1
python -m pytest tests -x -q -m gpu 2>&1 | tail -5
