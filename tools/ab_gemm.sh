#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests/test_gemm_gpu.py tests/test_fused_layers_gpu.py tests/test_model_gpu.py tests/test_transformer_gpu.py -x -q -m gpu 2>&1 | grep -E "^E  |passed|failed|Segm|rror" | head -12
for r in 1 2; do
for b in torch lt; do
  CODA_GEMM=$b python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$b', d['value'], d['ms_per_step'])"
done; done
python tools/cpu_bound.py 2>&1 | tail -4
