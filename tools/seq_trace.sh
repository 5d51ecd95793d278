R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/seqprof
CODA_BENCH_LEGS=headline rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/seqprof -o run -- python $R/bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 4 > /dev/null 2>&1
T=$(find $R/gpurun_out/seqprof -name run_kernel_trace.csv)
python $R/tools/trace_seq.py $T > $R/gpurun_out/seq.txt
rm -f $T
