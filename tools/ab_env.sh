#!/bin/bash
# dev: same-box A/B of one environment switch on the attention micro-benchmark:  ab_env.sh VAR [reps]
cd "$GRAFT_REPO_ROOT" || exit 1
VAR=$1; REPS=${2:-2}
python -m pytest tests/test_attention_gpu.py -x -q -m gpu 2>&1 | tail -2
for r in $(seq $REPS); do
  for v in 0 1; do echo "== $VAR=$v"; env $VAR=$v python tools/bench_attn.py; done
done
