#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_fused_bn_mlp_gpu.py tests/test_sa_module_gpu.py tests/test_syncbn_gpu.py -q -m gpu --timeout 600 > gpurun_out/e_tests.log 2>&1
tail -3 gpurun_out/e_tests.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/e_bench.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/e_bench.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print("headline", d["value"], d["ms_per_step"], "unchanged", d.get("value_unchanged"), d.get("ms_per_step_unchanged"), d["host"])
        print("roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
        for o in d.get("roofline_others", []):
            print("  %-60s frac %-8s ms %s rows %s" % (o["kernel"][:60], o["frac"], o.get("avg_launch_ms", o.get("sum_launch_ms")), o.get("packed_rows")))
        for k, v in d.get("extra_configs", {}).items():
            print(" ", k, v.get("value"), v.get("ms_per_step"))
            if "attention_kernels_bf16" in v:
                for kk, vv in v["attention_kernels_bf16"].items():
                    print("      ", kk, vv)
PY
