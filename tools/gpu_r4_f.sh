#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_sa_mfma_gpu.py tests/test_sa_module_gpu.py -q -m gpu --timeout 600 -x > gpurun_out/f_tests.log 2>&1
tail -5 gpurun_out/f_tests.log
timeout 300 python bench.py --workload sa --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/f_bench_sa.log 2>&1
timeout 900 python bench.py --no-cpu-baseline --no-extras > gpurun_out/f_bench.log 2>&1
python - <<'PY'
import json
for f in ("f_bench_sa", "f_bench"):
  for l in open(f"gpurun_out/{f}.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print(f, "headline", d["value"], d["ms_per_step"], "unchanged", d.get("value_unchanged"), d.get("ms_per_step_unchanged"), d["host"])
        for o in d.get("roofline_others", []):
            if "sa_" in o["kernel"] or "aggregate" in o["kernel"]:
                print("  %-60s frac %-8s ms %s" % (o["kernel"][:60], o["frac"], o.get("avg_launch_ms", o.get("sum_launch_ms"))))
PY
