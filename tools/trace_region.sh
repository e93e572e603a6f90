#!/bin/bash
# Kernel timeline (start offset, duration, stream, name) of ONE eager native train step at a region geometry:
#   tools/trace_region.sh 2x517x389   -> gpurun_out/tl_region_<shape>.txt
S=${1:-2x517x389}
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rm -rf gpurun_out/tlr_$S
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tlr_$S -o reg -- python3 tools/region_probe.py $S --steps 6 > gpurun_out/tlr_$S.log 2>&1
python3 - "$S" <<'PY'
import csv, glob, sys
s = sys.argv[1]
f = glob.glob(f"gpurun_out/tlr_{s}/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam_clip_fused" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["End_Timestamp"])
out = []
prev_end = t0
for r in rows[a + 1:b + 1]:
    st, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60]
    out.append(f"{(st - t0) / 1e3:9.1f} dur {(e - st) / 1e3:7.1f} q{r.get('Queue_Id', '?'):>3}  {name}")
open(f"gpurun_out/tl_region_{s}.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
rm -rf gpurun_out/tlr_$S
