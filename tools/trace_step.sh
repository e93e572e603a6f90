#!/bin/bash
# kernel timeline of one replayed step:  tools/trace_step.sh [bench args]  -> prints the last step's launches (start, duration)
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun (GRAFT_REPO_ROOT is the repository copy)}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/trace_step
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_step -o t -- python3 bench.py --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-extras --no-class-sweep "$@" > gpurun_out/trace_step.json 2> gpurun_out/trace_step.err
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace_step/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam_clip" in r["Kernel_Name"]]
a, b = idx[-2] + 1, idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f'{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} {r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]}')
PY
