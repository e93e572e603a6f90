#!/bin/bash
# per-kernel A/B of two library builds under rocprofv3 (same box): tools/ab_kernels.sh ab/libpopcorn_x.so [precision]
# -> average duration of every kernel of the replayed step with that library against the in-tree one
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
LIBA=$1; PREC=${2:-fp32}
for tag in a b; do
  rm -rf gpurun_out/abk_$tag
  if [ $tag = a ]; then export POPCORN_HIP_LIB=$LIBA; else unset POPCORN_HIP_LIB; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abk_$tag -o bench -- \
      python3 bench.py --precision $PREC --no-cpu-baseline --no-extras --no-class-sweep --no-config-legs --prewarm-seconds 1 > gpurun_out/abk_$tag.log 2>&1
done
python3 - <<'PY'
import csv, glob
def load(tag):
    f = glob.glob(f"gpurun_out/abk_{tag}/**/*kernel_stats.csv", recursive=True)[0]
    return {r["Name"]: (float(r["AverageNs"]) / 1e3, int(r["Calls"])) for r in csv.DictReader(open(f))}
a, b = load("a"), load("b")
steps = max(c for n, (t, c) in b.items() if "adam" in n)
rows = []
for n in b:
    if n in a and b[n][1] >= steps:
        k = b[n][1] / steps
        rows.append((k * (a[n][0] - b[n][0]), n, a[n][0], b[n][0], k))
rows.sort()
print("delta_us_per_step  A_us  in-tree_us  launches  kernel")
for d, n, ta, tb, k in rows:
    print(f"{d:8.1f} {ta:8.1f} {tb:8.1f} {k:5.1f}  {n[:100]}")
print("sum", sum(r[0] for r in rows))
PY
