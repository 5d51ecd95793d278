#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_attention_gpu.py -q -m gpu --timeout 600 > gpurun_out/j_tests.log 2>&1
tail -8 gpurun_out/j_tests.log
for ds in 1 0; do
CODA_ATTN_DS=$ds timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/j_bench_$ds.log 2>&1
done
python - <<'PY'
import json
for f in ("j_bench_1", "j_bench_0"):
  for l in open(f"gpurun_out/{f}.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print(f, "headline", d["value"], d["ms_per_step"], "unchanged", d.get("value_unchanged"), "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"].get("frac_whole_backward_8d"))
        for o in d.get("roofline_others", []):
            if "2048 x keys 2048" in o["kernel"]:
                print("  %-60s frac %-8s ms %s" % (o["kernel"][:60], o["frac"], o.get("avg_launch_ms")))
        print("  fps:", d["kernels_ms"])
PY
