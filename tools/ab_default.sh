#!/bin/bash
# dev: the default invocation (what the driver runs) twice on a fresh box: headline, value_unchanged with its two readings
# (right behind the headline leg / at the end of the run), host figures, wall time of the whole command
cd "$GRAFT_REPO_ROOT"
for r in 1 2; do
  t0=$(date +%s.%N)
  python bench.py 2>/dev/null > gpurun_out/ab_default_line$r.json
  t1=$(date +%s.%N)
  python - "$r" "$t0" "$t1" <<'PY'
import json, sys, glob, os
r, t0, t1 = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
d = json.loads(open(f"gpurun_out/ab_default_line{r}.json").read().strip().splitlines()[-1])
full = None
for f in ("gpurun_out/bench_full.json", "bench_full.json"):
    if os.path.exists(f):
        full = json.loads(open(f).read().strip().splitlines()[-1])
        break
print(f"run {r}: {t1 - t0:.1f} s wall; value", d["value"], "ms", d["ms_per_step"], "value_unchanged", d.get("value_unchanged"),
      "readings", (full or {}).get("value_unchanged_caller", {}).get("readings"), "host", d["host"], "line bytes",
      os.path.getsize(f"gpurun_out/ab_default_line{r}.json"))
PY
done
