#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_sa_mfma_gpu.py tests/test_sa_module_gpu.py -q -m gpu --timeout 600 -x > gpurun_out/h_tests.log 2>&1
tail -3 gpurun_out/h_tests.log
for occ in 1 2; do
CODA_SA_OCC=$occ timeout 300 python bench.py --workload sa --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/h_bench_sa_$occ.log 2>&1
CODA_SA_OCC=$occ timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/h_bench_$occ.log 2>&1
done
python - <<'PY'
import json
for f in ("h_bench_sa_1", "h_bench_sa_2", "h_bench_1", "h_bench_2"):
  for l in open(f"gpurun_out/{f}.log"):
    if l.startswith("{"):
        d = json.loads(l); print(f, d["value"], d["ms_per_step"])
        for o in d.get("roofline_others", []):
            if "sa_fwd" in o["kernel"] or "sa_mlp_agg" in o["kernel"]:
                print("  %-60s frac %-8s ms %s" % (o["kernel"][:60], o["frac"], o.get("avg_launch_ms", o.get("sum_launch_ms"))))
PY
