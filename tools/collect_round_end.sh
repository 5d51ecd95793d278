#!/bin/bash
# dev: copies what tools/run_round_end.sh left under gpurun_out/final_* into profiles/<tag>_* (run here, after the gpurun call)
cd "$(dirname "$0")/.." || exit 1
T=${1:-r06}; G=gpurun_out; P=profiles
cp $G/final_bench.json $P/${T}_bench_line.json
cp $G/final_bench_with_extras_full.json $P/${T}_bench_with_extras.json
{ tail -3 $G/final_tests.log; tail -1 $G/final_smoke.log; } > $P/${T}_gpu_tests_tail.txt
cp $G/final_bench_noprefetch.json $P/${T}_bench_noprefetch.json
cp $G/final_bench_sa.json $P/${T}_bench_sa.json
cp $G/final_bench_force_ddp.json $P/${T}_bench_force_ddp.json
cp $G/final_prof_bench.json $P/${T}_bench_under_rocprof.json
cp $G/final_prof/run_kernel_stats.csv $P/${T}_model_kernel_stats.csv
cp $G/final_prof/summary.md $P/${T}_model_summary.md
cp $G/final_prof/attention_by_grid.csv $P/${T}_attention_by_shape.csv
cp $G/final_prof/all_by_grid.csv $P/${T}_kernels_by_grid.csv
cp $G/final_bench_gemm_x3.txt $P/${T}_bench_gemm_x3.txt
cp $G/final_x3_bias.txt $P/${T}_x3_bias.txt; cp $G/final_x3_error.txt $P/${T}_x3_error.txt
if [ -z "$SKIP_PMC" ]; then
  cp $G/pmc_sa.txt $P/${T}_pmc_sa_mlp.md; cp $G/pmc_x3.txt $P/${T}_pmc_gemm_x3.md
  cp $G/final_pmc_attn_hbm.txt $P/${T}_pmc_attention_hbm.md; cp $G/final_pmc_attn_mfma.txt $P/${T}_pmc_attention_mfma.md
fi
git status --short $P | head -30
