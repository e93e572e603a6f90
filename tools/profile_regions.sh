#!/bin/bash
# Per-kernel time of the eager fused train step at the reference's REAL region geometry (weak_batch_size 2, variable-size census regions),
# one rocprofv3 --kernel-trace --stats run per shape:
#   tools/profile_regions.sh <tag> [shape ...]    -> gpurun_out/<tag>_regions_<shape>_kernel_stats.csv + <tag>_regions_<shape>.json
TAG=${1:-r5}; shift
SHAPES=${@:-2x230x220 2x700x640 2x1030x770}
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out
mkdir -p $OUT
for s in $SHAPES; do
  rm -rf $OUT/prof_reg_$s
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_reg_$s -o reg -- \
      python3 tools/region_probe.py $s --steps 10 --out $OUT/${TAG}_regions_$s.json > $OUT/${TAG}_regions_$s.log 2>&1
  f=$(find $OUT/prof_reg_$s -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${TAG}_regions_${s}_kernel_stats.csv
  rm -rf $OUT/prof_reg_$s
done
