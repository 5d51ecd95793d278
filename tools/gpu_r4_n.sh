#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf $R/gpurun_out/n_prof
CODA_BENCH_LEGS=headline rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/n_prof -o run -- \
  python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $R/gpurun_out/n_prof_bench.json 2>/dev/null
T=$(find $R/gpurun_out/n_prof -name run_kernel_trace.csv)
head -1 $T
python $R/tools/trace_gaps.py $T 0.5 > $R/gpurun_out/n_gaps.txt
cat $R/gpurun_out/n_gaps.txt
python $R/tools/prof_summary.py $(find $R/gpurun_out/n_prof -name run_kernel_stats.csv) 25 > $R/gpurun_out/n_summary.txt
head -24 $R/gpurun_out/n_summary.txt
gzip -c $T > $R/gpurun_out/n_trace.csv.gz
rm $T
