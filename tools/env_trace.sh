#!/bin/bash
# dev: kernel trace of the headline leg under two values of one environment switch:  env_trace.sh VAR VAL_A VAL_B [pattern]
# -> per (kernel, grid) averages matching `pattern`, the last step's span / busy time, and the per-category summary
R=$GRAFT_REPO_ROOT
VAR=$1; shift
PAT=${3:-.}
cd /tmp && export TMPDIR=/tmp
for v in "$1" "$2"; do
  D=$R/gpurun_out/envprof_$v
  rm -rf $D
  env $VAR=$v CODA_BENCH_LEGS=headline rocprofv3 --kernel-trace --stats --output-format csv -d $D -o run -- python $R/bench.py --no-cpu-baseline --no-extras --steps 12 --warmup 6 > $D.json 2>/dev/null
  T=$(find $D -name run_kernel_trace.csv)
  python $R/tools/trace_seq.py $T > $R/gpurun_out/env_seq_$v.txt
  python $R/tools/trace_by_grid.py $T _kernel > $R/gpurun_out/env_by_grid_$v.csv
  python $R/tools/prof_summary.py $(find $D -name run_kernel_stats.csv) auto > $R/gpurun_out/env_summary_$v.md
  rm -f $T
  echo "== $VAR=$v"; head -22 $R/gpurun_out/env_summary_$v.md; grep -E "$PAT" $R/gpurun_out/env_by_grid_$v.csv | head -40
  tail -1 $R/gpurun_out/env_seq_$v.txt
done
