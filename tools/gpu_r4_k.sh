#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_attention_gpu.py -q -m gpu -k "workspace or match_torch" > gpurun_out/k_tests.log 2>&1; tail -2 gpurun_out/k_tests.log
CODA_ATTN_DQG_KC=128 timeout 300 python -m pytest tests/test_attention_gpu.py -q -m gpu -k "workspace" > gpurun_out/k_tests2.log 2>&1; tail -2 gpurun_out/k_tests2.log
for tag in 64 128; do
CODA_ATTN_DQG_KC=$tag timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/k_bench_$tag.log 2>&1
done
python - <<'PY'
import json
for f in ("k_bench_64", "k_bench_128"):
  for l in open(f"gpurun_out/{f}.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print(f, "headline", d["value"], d["ms_per_step"], "unchanged", d.get("value_unchanged"), "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
        for o in d.get("roofline_others", []):
            if "dq_gemm" in o["kernel"]: print("   dq gemm", o["frac"], o["avg_launch_ms"])
PY
