#!/bin/bash
# A/B of an environment switch on the bench line:  tools/ab_env.sh VAR "bench args"  -> alternates VAR=0 / VAR=1 twice
VAR=$1; shift
for v in 0 1 0 1; do
  env $VAR=$v python bench.py "$@" --no-class-sweep --no-cpu-baseline --no-extras --no-config-legs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$VAR=$v', d['value'], d['ms_per_step'], d['final_loss'])"
done
