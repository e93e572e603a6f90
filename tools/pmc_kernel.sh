#!/bin/bash
# SQ counters of one kernel of the step (eager dispatches):  tools/pmc_kernel.sh <kernel-name-substring> [bench args]
# separate --pmc passes with --kernel-trace only; prints the mean per launch of every counter for the matching kernel
K=$1; shift
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the repository copy)}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_FLAT" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  rm -rf gpurun_out/pmck_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d gpurun_out/pmck_$i -o k -- \
      python3 bench.py --steps 2 --warmup 1 --repeats 1 --no-graph --no-cpu-baseline --no-extras --no-class-sweep --no-config-legs --prewarm-seconds 0 "$@" > gpurun_out/pmck_$i.log 2>&1
done
python3 - "$K" <<'PY'
import collections, csv, glob, sys
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmck_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[1] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:32s} n={len(v):3d} mean {sum(v) / len(v):16.1f}")
PY
