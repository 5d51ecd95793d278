#!/bin/bash
# round 4, first GPU call: the new MFMA pipeline's unit tests, the SA module tests, a short bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_sa_mfma_gpu.py tests/test_sa_module_gpu.py -q -m gpu -x --timeout 300 > gpurun_out/a_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/a_tests.log
tail -30 gpurun_out/a_tests.log
timeout 300 python bench.py --workload sa --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/a_bench_sa.log 2>&1
tail -c 1500 gpurun_out/a_bench_sa.log
CODA_SA_MLP=fused timeout 300 python bench.py --workload sa --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/a_bench_sa_old.log 2>&1
tail -c 600 gpurun_out/a_bench_sa_old.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/a_bench.log 2>&1
tail -c 3000 gpurun_out/a_bench.log
