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
import json, os
if os.environ.get("PMCK_JSON"):
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    d = {"kernel_substring": sys.argv[1], "launches_per_counter": {k: len(v) for k, v in acc.items()}, "mean_per_launch": m}
    # SQ counters are summed over the chip: ratios are what is comparable between kernels
    if m.get("SQ_WAVE_CYCLES"):
        d["wait_inst_any_over_wave_cycles"] = m.get("SQ_WAIT_INST_ANY", 0.0) / m["SQ_WAVE_CYCLES"]
    if m.get("SQ_BUSY_CYCLES") and m.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
        # SQ_BUSY_CYCLES counts per SE-level SQ; MFMA-busy per SIMD: normalise by the kernel's own duration in GRBM cycles over 1024 SIMDs
        if m.get("GRBM_GUI_ACTIVE"):
            d["mfma_busy_per_simd_over_kernel_cycles"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (m["GRBM_GUI_ACTIVE"] / 8.0)
    d["note"] = os.environ.get("PMCK_NOTE", "")
    json.dump(d, open(os.environ["PMCK_JSON"], "w"), indent=1)
PY
