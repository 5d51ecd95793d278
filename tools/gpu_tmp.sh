#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/run_variant.py scripts 2>&1 | tail -2
export TMPDIR=/tmp
cd /tmp
rm -rf $R/gpurun_out/s_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/s_prof -o run -- python $R/tools/run_variant.py scripts > /dev/null 2>&1
find $R/gpurun_out/s_prof -name run_kernel_trace.csv -delete
python $R/tools/prof_summary.py $(find $R/gpurun_out/s_prof -name run_kernel_stats.csv) 13 | head -60
