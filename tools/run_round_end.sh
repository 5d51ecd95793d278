#!/bin/bash
# dev: one GPU-box call that refreshes the round's measured artifacts under gpurun_out/final_* (copied by hand into
# profiles/ afterwards).  Usage: gpurun -- 'bash tools/run_round_end.sh [tests]'
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ "$1" = "tests" ]; then
  timeout 1500 python -m pytest tests -q -m gpu --timeout 900 > gpurun_out/final_tests.log 2>&1
  tail -3 gpurun_out/final_tests.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1; tail -2 gpurun_out/final_smoke.log
fi
python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
cp gpurun_out/bench_full.json gpurun_out/final_bench_with_extras_full.json
python bench.py --prefetch off --no-cpu-baseline --no-extras > gpurun_out/final_bench_noprefetch.json 2>> gpurun_out/final_bench.err
python bench.py --workload sa --no-cpu-baseline > gpurun_out/final_bench_sa.json 2>> gpurun_out/final_bench.err
# the multi-GPU wrapping forced onto the one rank (SyncBatchNorm conversion, RCCL group of 1, gradient all-reduce executed,
# comm diagnostics, the unchanged leg under DistributedDataParallel)
CODA_BENCH_FORCE_DDP=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 python bench.py --no-cpu-baseline --no-extras > gpurun_out/final_bench_force_ddp.json 2>> gpurun_out/final_bench.err
export TMPDIR=/tmp
cd /tmp
rm -rf $R/gpurun_out/final_prof
# the headline leg alone (CODA_BENCH_LEGS=headline: no unchanged-caller leg), so that per-step figures are the headline's
CODA_BENCH_LEGS=headline rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final_prof -o run -- \
  python $R/bench.py --no-cpu-baseline --no-extras > $R/gpurun_out/final_prof_bench.json 2>/dev/null
python $R/tools/trace_by_grid.py $(find $R/gpurun_out/final_prof -name run_kernel_trace.csv) > $R/gpurun_out/final_prof/attention_by_grid.csv
python $R/tools/trace_gaps.py $(find $R/gpurun_out/final_prof -name run_kernel_trace.csv) 0.3 > $R/gpurun_out/final_prof/gaps.txt
python $R/tools/prof_summary.py $(find $R/gpurun_out/final_prof -name run_kernel_stats.csv) auto > $R/gpurun_out/final_prof/summary.md
python $R/tools/trace_by_grid.py $(find $R/gpurun_out/final_prof -name run_kernel_trace.csv) _kernel > $R/gpurun_out/final_prof/all_by_grid.csv
find $R/gpurun_out/final_prof -name run_kernel_trace.csv -delete   # tens of MB; the stats file is what gets committed
cd $R
export PMC_TAG=r06
if [ -z "$SKIP_PMC" ]; then   # (SKIP_PMC=1: the counter passes, when the kernels they look at have not changed)
timeout 300 python tools/sa_prof.py > gpurun_out/final_sa_prof.txt 2>&1
bash tools/pmc_sa.sh > /dev/null 2>&1
bash tools/pmc_attn.sh > gpurun_out/final_pmc_attn_hbm.txt 2>&1
bash tools/pmc_attn_mfma.sh > gpurun_out/final_pmc_attn_mfma.txt 2>&1
bash tools/pmc_x3.sh > /dev/null 2>&1
fi
cp gpurun_out/bench_full.json gpurun_out/final_bench_full.json 2>/dev/null
python tools/bench_gemm_x3.py > gpurun_out/final_bench_gemm_x3.txt 2>&1
python tools/x3_bias.py > gpurun_out/final_x3_bias.txt 2>&1
python tools/x3_error.py > gpurun_out/final_x3_error.txt 2>&1
python - <<'PY'
import json
for f in ("final_bench", "final_bench_noprefetch", "final_bench_sa", "final_bench_force_ddp", "final_prof_bench"):
    for l in open(f"gpurun_out/{f}.json"):
        if not l.startswith("{"):
            continue
        d = json.loads(l)
        r = d["roofline"]
        print(f, d["value"], d["ms_per_step"], d.get("value_unchanged"), r["kernel"][:40], r["achieved"], r["frac"], r.get("avg_launch_ms"),
              d.get("cpu_baseline", {}).get("value"))
PY
