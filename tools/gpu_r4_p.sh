#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_sa_mfma_gpu.py tests/test_sa_module_gpu.py -q -m gpu -x 2>&1 | tail -5 > gpurun_out/p_tests.log; tail -5 gpurun_out/p_tests.log
timeout 300 python tools/sa_prof.py > gpurun_out/p_sa_prof.txt 2>&1; cat gpurun_out/p_sa_prof.txt
timeout 300 python bench.py --workload sa --no-cpu-baseline > gpurun_out/p_bench_sa.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/p_bench_sa.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print("sa", d["value"], d["ms_per_step"])
        for o in d.get("roofline_others", []):
            if "sa_" in o["kernel"] or "fps" in o["kernel"]: print("   %-70s %s %s" % (o["kernel"][:70], o["frac"], o.get("avg_launch_ms", o.get("sum_launch_ms"))))
PY
