#!/bin/bash
# Round 6: everything under profiles/r6_* in one GPU call (tools/profile_round.sh r6 + the before / after SQ passes of the fused conv backward +
# the region-size kernel stats).  Outputs land in gpurun_out/; copy what is to be judged into profiles/.
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/profile_round.sh r6 > gpurun_out/r6_profile_round.log 2>&1
# before: the fp32-MFMA form of the fused backward (pc_set_conv_split 0 = round 5's kernel); after: the split-operand kernel
export PMCK_JSON=gpurun_out/r6_pmc_conv_bwd_before.json PMCK_NOTE="BEFORE: conv3x3_bwd_f32_kernel (v_mfma_f32_16x16x4_f32; POPCORN_CONV_SPLIT=0), SQ counters per launch, mean over the step's launches; separate --pmc passes of an eager bench run"
POPCORN_CONV_SPLIT=0 bash tools/pmc_kernel.sh conv3x3_bwd_f32_kernel > gpurun_out/r6_pmc_conv_bwd_before.txt 2>&1
export PMCK_JSON=gpurun_out/r6_pmc_conv_bwd_after.json PMCK_NOTE="AFTER: conv3x3_bwd_s3_kernel<8, false> (operands split into three bf16 planes while staged, v_mfma_f32_16x16x32_bf16), SQ counters per launch, mean over the step's five launches; separate --pmc passes of an eager bench run"
bash tools/pmc_kernel.sh "conv3x3_bwd_s3_kernel<8, false>" > gpurun_out/r6_pmc_conv_bwd_after.txt 2>&1
# the same for the forward convs in split form (csrc/conv3x3_fwd_s3.h), on the largest of them: the first conv of up1 from the low-resolution map
export PMCK_JSON=gpurun_out/r6_pmc_conv_fwd_before.json PMCK_NOTE="BEFORE: conv3x3_mfma_kernel<8, 8, 0, 1, 0, 8> (composed Up conv 8 + 8z -> 8 @128x128 x4, v_mfma_f32_16x16x4_f32; POPCORN_FWD_SPLIT=0: only the forward kernels change form), SQ counters per launch; separate --pmc passes of an eager bench run"
POPCORN_FWD_SPLIT=0 bash tools/pmc_kernel.sh "conv3x3_mfma_kernel<8, 8, 0, 1, 0, 8>" > gpurun_out/r6_pmc_conv_fwd_before.txt 2>&1
export PMCK_JSON=gpurun_out/r6_pmc_conv_fwd_after.json PMCK_NOTE="AFTER: conv3x3_fwd_s3_kernel<8, 8, 0, 8> (same launch; operands split into three bf16 planes while staged, v_mfma_f32_16x16x32_bf16), SQ counters per launch; separate --pmc passes of an eager bench run"
bash tools/pmc_kernel.sh "conv3x3_fwd_s3_kernel<8, 8, 0, 8>" > gpurun_out/r6_pmc_conv_fwd_after.txt 2>&1
unset PMCK_JSON PMCK_NOTE
bash tools/profile_regions.sh r6 2x230x220 2x517x389 2x700x640 > gpurun_out/r6_profile_regions.log 2>&1
rm -rf gpurun_out/pmck_*
ls -la gpurun_out | grep r6_
