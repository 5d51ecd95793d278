#!/bin/bash
# dev: correctness + same-box A/B of one switch of the encoder-shape attention backward:  ab_dkv.sh [VAR]
# (CODA_ATTN_DKV_X3, CODA_ATTN_DQ_X3)
cd "$GRAFT_REPO_ROOT" || exit 1
VAR=${1:-CODA_ATTN_DKV_X3}
python -m pytest tests/test_attention_gpu.py -x -q -m gpu 2>&1 | tail -3
for r in 1 2; do for v in 0 1; do echo "== $VAR=$v"; env $VAR=$v python tools/bench_attn.py fp32 2>&1 | grep "L= 2048"; done; done
