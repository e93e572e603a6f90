#!/bin/bash
# Kernel timeline of one replayed train step (start offset, duration, name) from a short rocprofv3 --kernel-trace run of the bench:
#   tools/step_timeline.sh [tag] [bench args]     -> gpurun_out/r3/tl_<tag>.txt
TAG=${1:-a}; shift
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3
rm -rf gpurun_out/r3/tl_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3/tl_$TAG -o bench -- python3 bench.py --steps 10 --warmup 2 --repeats 2 --no-cpu-baseline --no-class-sweep --no-extras "$@" > gpurun_out/r3/tl_$TAG.json 2> gpurun_out/r3/tl_$TAG.err
python3 - "$TAG" <<'PY'
import csv, sys
tag = sys.argv[1]
rows = list(csv.DictReader(open(f"gpurun_out/r3/tl_{tag}/bench_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam_clip_fused" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["End_Timestamp"])
out = []
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:70]
    out.append(f"{(s - t0) / 1e3:9.1f} dur {(e - s) / 1e3:7.1f}  {name}")
open(f"gpurun_out/r3/tl_{tag}.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
