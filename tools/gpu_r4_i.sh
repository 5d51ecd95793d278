#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/op_sites.py > gpurun_out/i_op_sites.log 2>&1
export TMPDIR=/tmp
cd /tmp
rm -rf $R/gpurun_out/i_prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/i_prof -o run -- \
  python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $R/gpurun_out/i_prof_bench.json 2>/dev/null
find $R/gpurun_out/i_prof -name run_kernel_trace.csv -delete
cd $R
python tools/prof_summary.py $(find gpurun_out/i_prof -name run_kernel_stats.csv) 36 --md | head -30
