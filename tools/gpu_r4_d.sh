#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_attention_bf16_gpu.py tests/test_gemm_gpu.py tests/test_call_options.py "tests/test_full_step_gpu.py::test_whole_step_forward_criterion_backward[configs4_40k_512q_bf16]" -q -m gpu -s --timeout 600 > gpurun_out/d_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/d_tests.log
grep -E "bf16 kernels vs|passed|failed|configs4" gpurun_out/d_tests.log | tail -20
timeout 300 python tools/host_time_small.py > gpurun_out/d_host_time.log 2>&1; tail -3 gpurun_out/d_host_time.log
timeout 300 python tools/gemm_shapes.py > gpurun_out/d_gemm_shapes.log 2>&1; tail -5 gpurun_out/d_gemm_shapes.log
timeout 300 python tools/host_profile.py > gpurun_out/d_host_profile.log 2>&1; head -5 gpurun_out/d_host_profile.log
