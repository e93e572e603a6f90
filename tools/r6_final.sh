#!/bin/bash
# Round 6, final collection in one GPU call: profiles (tools/r6_profiles.sh) copied into profiles/, THEN the two bench lines (they quote the
# stamped profile), then the GPU suite.
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/r6_profiles.sh > gpurun_out/r6_profiles_run.log 2>&1
for f in r6_fp32_graph_kernel_stats.csv r6_bf16_graph_kernel_stats.csv r6_stamp.json r6_pmc_conv_8to8.json r6_pmc_conv_8to8_bf16.json r6_pmc_conv_bwd.json \
         r6_pmc_conv_bwd_after.json r6_pmc_conv_bwd_before.json r6_pmc_conv_fwd_after.json r6_pmc_conv_fwd_before.json r6_pmc_head_bwd.json r6_pmc_head_bwd_bf16.json r6_pmc_step_kernels.json \
         r6_pmc_step_kernels_bf16.json r6_regions_2x230x220_kernel_stats.csv r6_regions_2x517x389_kernel_stats.csv r6_regions_2x700x640_kernel_stats.csv; do
  cp gpurun_out/$f profiles/$f
done
python3 bench.py > gpurun_out/r6_bench_final.json 2> gpurun_out/r6_bench_final.err
python3 bench.py --precision bf16 > gpurun_out/r6_bench_final_bf16.json 2> gpurun_out/r6_bench_final_bf16.err
wc -c gpurun_out/r6_bench_final.json gpurun_out/r6_bench_final_bf16.json
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4
