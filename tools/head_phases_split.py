"""Phase profile of the split-operand producer waves of head_bwd_pc_kernel<0, true> (fp32 mode, B = 64, 100 x 100, every pixel selected)
and the role ablation of the kernel in both product forms.  Needs the profiling build:

    tools/build_variant.sh prof -DPOPCORN_HEAD_PROF                                             (here)
    gpurun -- 'python tools/head_phases_split.py --out gpurun_out/r5_head_bwd_split_phases.json'

The stamps (s_memtime into scalar accumulators, HR_CLOSE in head.hip) wait for outstanding LDS operations, i.e. they serialise the phases:
read the numbers as upper bounds of each phase, their sum as the stamped build's group time.  GPU only."""
import argparse
import json
import os
import re
import subprocess
import sys

NAMES = ["loop_top", "forward_chain", "output_layer_g3", "wait_slot0", "write_slot0", "dgrad3", "wait_slot1", "write_slot1", "dgrad2",
         "wait_slot2", "write_slot2", "feature_gradient_store"]


def ablate(env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "tools/ablate_head.py"] + (["--phases"] if env.get("POPCORN_HEAD_PROF") else []), env=e,
                       capture_output=True, text=True)
    return r.stdout


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/r5_head_bwd_split_phases.json")
    ap.add_argument("--lib", default="ab/libpopcorn_prof.so")
    a = ap.parse_args()
    res = {"kernel": "head_bwd_pc_kernel<0, true>", "workload": "B = 64, 100 x 100 crop of 128 x 128 feature maps, every pixel selected",
           "call_us": {}}
    for form, env in (("split", {}), ("fp32_mfma", {"POPCORN_HEAD_SPLIT": "0"})):
        out = ablate(env)
        d = {}
        for line in out.splitlines():
            m = re.match(r"(pc full|single-role|consumer idle|no hand-off)\s+([0-9.]+) us", line)
            if m:
                d[m.group(1).replace(" ", "_").replace("-", "_")] = float(m.group(2))
        res["call_us"][form] = d
    res["call_us"]["note"] = ("pack + kernel + reduce launches of one pc_head_bwd call, 10 back-to-back calls (tools/ablate_head.py); "
                              "single-role = the fp32-MFMA one-role kernel in both rows")
    if os.path.exists(a.lib):
        out = ablate({"POPCORN_HIP_LIB": a.lib, "POPCORN_HEAD_PROF": "1"})
        for line in out.splitlines():
            m = re.match(r"(pc full|consumer idle|no hand-off)\s+([0-9. ]+)$", line)
            if m:
                nums = [float(x) for x in m.group(2).split()]
                if len(nums) == 12:
                    res.setdefault("producer_cycles_per_group", {})[m.group(1).replace(" ", "_").replace("-", "_")] = dict(zip(NAMES, nums))
        full = res.get("producer_cycles_per_group", {}).get("pc_full")
        if full:
            res["sum_cycles_per_group"] = sum(full.values())
    else:
        res["error_prof"] = f"{a.lib} missing: tools/build_variant.sh prof -DPOPCORN_HEAD_PROF"
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
