#!/bin/bash
# round 4: kernel trace of the SA-only workload (new MFMA pipeline) and of the full step
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf $R/gpurun_out/b_prof_sa $R/gpurun_out/b_prof_model
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/b_prof_sa -o run -- \
  python $R/bench.py --workload sa --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/b_prof_sa.json 2>/dev/null
rm -f $R/gpurun_out/b_prof_sa/*/run_kernel_trace.csv $R/gpurun_out/b_prof_sa/run_kernel_trace.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/b_prof_model -o run -- \
  python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $R/gpurun_out/b_prof_model.json 2>/dev/null
python $R/tools/trace_by_grid.py $(find $R/gpurun_out/b_prof_model -name run_kernel_trace.csv) > $R/gpurun_out/b_prof_model/attention_by_grid.csv
find $R/gpurun_out/b_prof_model -name run_kernel_trace.csv -delete
cd $R
find gpurun_out/b_prof_sa gpurun_out/b_prof_model -type f | head
python tools/prof_summary.py $(find gpurun_out/b_prof_sa -name run_kernel_stats.csv) 13 --md | head -60
