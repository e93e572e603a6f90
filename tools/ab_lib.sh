#!/bin/bash
# A/B of two builds of the library on the bench line (same box, alternating):  tools/ab_lib.sh ab/libpopcorn_base.so [bench args]
# -> POPCORN_HIP_LIB=<that file> against the in-tree library, twice each; prints value, ms_per_step, the head backward's live time, loss
BASE=$1; shift
for lib in "$BASE" "" "$BASE" ""; do
  env POPCORN_HIP_LIB=$lib python bench.py "$@" --no-class-sweep --no-cpu-baseline --no-extras --no-config-legs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('${lib:-in-tree}', d['value'], d['ms_per_step'], d['roofline']['launch_us'], d['final_loss'])"
done
