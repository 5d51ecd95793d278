#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_sa_mfma_gpu.py tests/test_sa_module_gpu.py -q -m gpu -x 2>&1 | tail -5 > gpurun_out/q_tests.log; tail -5 gpurun_out/q_tests.log
timeout 300 python tools/sa_prof.py > gpurun_out/q_sa_prof.txt 2>&1; grep -A2 "fwd layer 3" gpurun_out/q_sa_prof.txt
for mode in fused; do
CODA_SA_POOL=$mode timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/q_bench_$mode.log 2>&1
done
python - <<'PY'
import json
for f in ("q_bench_fused",):
  for l in open(f"gpurun_out/{f}.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print(f, "headline", d["value"], d["ms_per_step"], "unchanged", d.get("value_unchanged"))
        for o in d.get("roofline_others", []):
            if "sa_" in o["kernel"] or "pool_rows" in o["kernel"]: print("   %-60s %s %s" % (o["kernel"][:60], o["frac"], o.get("avg_launch_ms", o.get("sum_launch_ms"))))
PY
