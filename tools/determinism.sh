#!/bin/bash
# run-to-run determinism of the captured step: the loss after a FIXED number of steps (no time-based pre-warm), N runs
N=${1:-6}; shift
for i in $(seq $N); do
  python bench.py --prewarm-seconds 0 --steps 200 --repeats 1 --no-class-sweep --no-cpu-baseline --no-extras --no-config-legs "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['ms_per_step'], d['final_loss'], d.get('prewarm_steps'))"
done
