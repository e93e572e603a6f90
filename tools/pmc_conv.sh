#!/bin/bash
# PMC passes over the grouped conv ablation (tools/ablate_conv_group.py); usage: tools/pmc_conv.sh CIN COUT HW [dbg]
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the repository copy)}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
export ABL_ONE=${4:-0}
i=0
for pmc in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum" \
           "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_64B_sum TCC_CYCLE_sum"; do
  i=$((i+1))
  # (a pass with TA_* / TCP_PENDING_* counters aborted the profiler and hung the call on this pool: not collected)
  timeout 180 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d gpurun_out/pmc_c$i -o c -- python3 tools/ablate_conv_group.py $1 $2 $3 > gpurun_out/pmc_c$i.log 2>&1
  python3 tools/pmc_summary.py gpurun_out/pmc_c$i/c_counter_collection.csv | grep -A6 conv3x3 | grep -v "^at::"
done
