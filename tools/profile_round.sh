#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel stats of the bench (fp32 and bf16 lines, graph replay) and the FETCH_SIZE /
# WRITE_SIZE PMC passes (separate runs, --kernel-trace only, eager dispatches) that feed `roofline.traffic`, for both modes.
#   usage: tools/profile_round.sh r4 [quick]   -> gpurun_out/<tag>_*  (copy what is to be judged into profiles/)
#   quick: only the two kernel-stats runs + the source stamp (refresh after a kernel commit); full: + all PMC passes
TAG=${1:-r6}
MODE=${2:-full}
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the repository copy)}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out
mkdir -p $OUT
for prec in fp32 bf16; do
  rm -rf $OUT/prof_$prec
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$prec -o bench -- \
      python3 bench.py --steps 10 --warmup 2 --repeats 3 --no-cpu-baseline --no-class-sweep --no-extras --no-config-legs --precision $prec > $OUT/${TAG}_${prec}_bench_under_profiler.json 2> $OUT/prof_$prec.err
  f=$(find $OUT/prof_$prec -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${TAG}_${prec}_graph_kernel_stats.csv
  [ "$MODE" = quick ] && continue
  # PMC: one counter set per run (FETCH_SIZE costs 3 of the 4 TCC slots), eager launches so that every kernel is a dispatch
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/pmc_${prec}_$c
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${prec}_$c -o b -- \
        python3 bench.py --steps 2 --warmup 1 --repeats 1 --prewarm-seconds 0 --no-graph --no-cpu-baseline --no-class-sweep --no-extras --no-config-legs --precision $prec > $OUT/pmc_${prec}_$c.log 2>&1
  done
done
# the stamp: which kernel sources these summaries belong to (tests/test_bench_contract.py compares it with the tree)
python3 tools/csrc_hash.py $TAG > $OUT/${TAG}_stamp.json
if [ "$MODE" = quick ]; then rm -rf $OUT/prof_fp32 $OUT/prof_bf16; exit 0; fi
# the standalone grouped conv 8->8 @128x128 x4 launch (roofline_conv): the precision goes through exported variables (ABL_PREC) that
# the profiled python program reads itself -- the program after `--` is python3, never `env`
for prec in fp32 bf16; do
  for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
    n=$(echo $c | cut -d' ' -f1)
    rm -rf $OUT/pmcc_${prec}_$n
    export ABL_ONE=0 ABL_PREC=$prec
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmcc_${prec}_$n -o c -- \
        python3 tools/ablate_conv_group.py 8 8 128 > $OUT/pmcc_${prec}_$n.log 2>&1
  done
done
python3 tools/pmc_round_summary.py $TAG
# only the summaries travel back (gpurun merges at most 64 MiB of gpurun_out/)
rm -rf $OUT/prof_fp32 $OUT/prof_bf16 $OUT/pmc_fp32_* $OUT/pmc_bf16_* $OUT/pmcc_fp32_* $OUT/pmcc_bf16_*
