#!/bin/bash
# dev: one GPU-box call that refreshes the round's measured artifacts under gpurun_out/
#   (copied by hand into profiles/ afterwards).  Usage: gpurun -- 'bash tools/run_round_end.sh [tests]'
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
if [ "$1" = "tests" ]; then
  python -m pytest tests -x -q -m gpu 2>&1 | tail -3
fi
python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
python bench.py --prefetch off --no-cpu-baseline > gpurun_out/final_bench_noprefetch.json 2>> gpurun_out/final_bench.err
python bench.py --workload sa --no-cpu-baseline > gpurun_out/final_bench_sa.json 2>> gpurun_out/final_bench.err
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf $R/gpurun_out/final_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final_prof -o run -- \
  python $R/bench.py --no-cpu-baseline --no-extras > $R/gpurun_out/final_prof_bench.json 2>/dev/null
python $R/tools/trace_by_grid.py $R/gpurun_out/final_prof/run_kernel_trace.csv > $R/gpurun_out/final_prof/attention_by_grid.csv
rm -f $R/gpurun_out/final_prof/run_kernel_trace.csv   # tens of MB; the stats file is what gets committed
rm -rf $R/gpurun_out/final_tower_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final_tower_prof -o tower -- \
  python $R/tools/bench_clip_tower.py --iters 30 > $R/gpurun_out/final_tower.json 2>/dev/null
rm -f $R/gpurun_out/final_tower_prof/tower_kernel_trace.csv
cd $R
cat gpurun_out/final_tower.json
python - <<'PY'
import json
for f in ("final_bench", "final_bench_noprefetch", "final_bench_sa", "final_prof_bench"):
    d = json.load(open(f"gpurun_out/{f}.json"))
    r = d["roofline"]
    print(f, d["value"], d["ms_per_step"], r["kernel"][:40], r["achieved"], r["frac"], r.get("avg_launch_ms"),
          d.get("cpu_baseline", {}).get("value"))
PY
