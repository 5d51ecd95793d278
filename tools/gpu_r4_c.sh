#!/bin/bash
# round 4: full GPU test suite + SA-only bench + headline bench (short)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu --timeout 600 > gpurun_out/c_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/c_tests.log
tail -15 gpurun_out/c_tests.log
timeout 300 python bench.py --workload sa --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/c_bench_sa.log 2>&1
tail -c 400 gpurun_out/c_bench_sa.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/c_bench.log 2>&1
python - <<'PY'
import json
for f in ("c_bench_sa", "c_bench"):
    for l in open(f"gpurun_out/{f}.log"):
        if l.startswith("{"):
            d = json.loads(l)
            print(f, d["value"], d["ms_per_step"], d.get("value_unchanged"), d.get("ms_per_step_unchanged"), d["host"])
PY
