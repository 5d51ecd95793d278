#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_attention_gpu.py -q -m gpu > gpurun_out/k_tests.log 2>&1; tail -4 gpurun_out/k_tests.log
for tag in 1; do
CODA_ATTN_DS=$tag timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/k_bench_$tag.log 2>&1
done
python - <<'PY'
import json
for f in ("k_bench_1", "k_bench_0"):
  for l in open(f"gpurun_out/{f}.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print(f, "headline", d["value"], d["ms_per_step"], "unchanged", d.get("value_unchanged"), "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
        for o in d.get("roofline_others", []):
            if "mha" in o["kernel"] or "decoder_agg" in o["kernel"]: print("   %-70s %s %s" % (o["kernel"][:70], o["frac"], o.get("avg_launch_ms", o.get("sum_launch_ms"))))
PY
