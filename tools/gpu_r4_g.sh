#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_sa_mfma_gpu.py tests/test_sa_module_gpu.py -q -m gpu --timeout 600 -x > gpurun_out/g_tests.log 2>&1
tail -3 gpurun_out/g_tests.log
timeout 300 python bench.py --workload sa --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/g_bench_sa.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/g_bench_sa.log"):
    if l.startswith("{"):
        d = json.loads(l); print("sa-only", d["value"], d["ms_per_step"])
PY
bash tools/pmc_sa.sh | grep -A2 "^sa_"
