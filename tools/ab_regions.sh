#!/bin/bash
# A/B of an environment variable on the config3_regions leg: tools/ab_regions.sh VAR valA valB  -> gpurun_out/ab_regions_<VAR>.txt
VAR=$1; A=$2; B=$3
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
OUT=gpurun_out/ab_regions_$VAR.txt
: > $OUT
for rep in 1 2; do
  for v in $A $B; do
    export $VAR=$v
    python3 tools/bench_regions.py 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)
print('$VAR=$v', d['step_tflops'], ' '.join('%s:%.3f' % (r['batch'], r['ms_per_step']) for r in d['batches']))" >> $OUT
  done
done
cat $OUT
