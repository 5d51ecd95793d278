#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for tag in 1 0; do
CODA_ATTN_DQG_NT=$tag timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/m_bench_$tag.log 2>&1
done
python - <<'PY'
import json
for f in ("m_bench_1", "m_bench_0"):
  for l in open(f"gpurun_out/{f}.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print(f, "headline", d["value"], d["ms_per_step"], "unchanged", d.get("value_unchanged"), "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
        for o in d.get("roofline_others", []):
            if "dq_gemm" in o["kernel"]: print("   dq gemm", o["frac"], o["avg_launch_ms"])
PY
