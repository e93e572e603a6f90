"""Phase profile of head_bwd_bf16_coop4_kernel (B = 64, 100 x 100, every pixel selected = the bench step's launch): s_memtime stamps
around the phases of an iteration (loop top / forward chain / backward chain + store / exchange writes / barrier 1 / weight gradients /
barrier 2), summed per wave over the kernel, averaged over all waves.  Needs the profiling build of the library:

    tools/build_variant.sh prof -DPOPCORN_HEAD_PROF            (here)
    gpurun -- 'python tools/head_phases_bf16.py --out gpurun_out/r5_head_bwd_bf16_phases.json'

Also times the product library's launch (events over 20 back-to-back calls on rotating feature maps) so that the stamped build's
distortion is visible.  GPU only."""
import argparse
import json
import os
import re
import subprocess
import sys

CODE = r'''
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from popcorn_amd import ops, _lib as L
from popcorn_amd.model import POPCORN
torch.manual_seed(0)
m = POPCORN(6, occupancymodel=True, pretrained=True, biasinit=0.9, sentinelbuildings=True).cuda()
m.set_precision("bf16")
B, H, W = 64, 100, 100
feats = [torch.randn(B, 16, 128, 128, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(4)]
building = torch.rand(B, 1, H, W, device="cuda")
admin = torch.ones(B, H, W, device="cuda"); census = torch.ones(B, dtype=torch.int64, device="cuda")
gpc = torch.ones(B, device="cuda"); gsc = torch.full((1,), 1e-3, device="cuda")
grads = [torch.empty_like(t) for t in m.head_tensors()]
eng = m.engines()[0]
with L.precision("bf16"):
    gf = L.empty_act(B, 16, 128, 128, feats[0].device)
    def run(i):
        ops.head_bwd(feats[i % 4], 14, 14, H, W, m.head_tensors(), building, admin_mask=admin, census_idx=census, g_popcount=gpc,
                     g_scale_const=gsc, grads=grads, g_feat=gf, feat_bn=eng.feat_bn())
    for i in range(3): run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(20): run(i)
    e1.record(); torch.cuda.synchronize()
print("CALL_US %.2f" % (e0.elapsed_time(e1) * 50))
'''


def one(env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", CODE], env=e, capture_output=True, text=True)
    m = re.search(r"CALL_US ([0-9.]+)", r.stdout)
    ph = [l for l in r.stderr.splitlines() if "head_bwd_bf16_coop4 phases" in l]
    return (float(m.group(1)) if m else None), (ph[-1] if ph else None), r.stderr[-600:]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/r5_head_bwd_bf16_phases.json")
    ap.add_argument("--lib", default="ab/libpopcorn_prof.so")
    a = ap.parse_args()
    us, _, err = one({})
    res = {"kernel": "head_bwd_bf16_coop4_kernel", "workload": "B = 64, 100 x 100 crop of a 128 x 128 channels-last bf16 feature map, every pixel selected",
           "product_call_us": us, "note_call": "pack + kernel + reduce launches of one pc_head_bwd call, 20 back-to-back calls on 4 rotating feature maps"}
    if us is None:
        res["error"] = err
    if os.path.exists(a.lib):
        us_p, line, err = one({"POPCORN_HIP_LIB": a.lib, "POPCORN_HEAD_PROF": "1"})
        res["stamped_call_us_incl_sync_and_readback"] = us_p
        if line:
            nums = [float(x) for x in line.split("):")[-1].split()]
            names = ["loop_top", "forward_chain", "backward_chain_store", "exchange_writes", "barrier1", "weight_gradients", "barrier2"]
            mwg = re.search(r"(\d+) workgroups, (\d+) iterations", line)
            res["cycles_per_iteration_per_wave"] = dict(zip(names, nums[:7]))
            tot = sum(nums[:7])
            res["sum_cycles_per_iteration"] = tot
            res["share"] = {k: round(v / tot, 4) for k, v in zip(names, nums[:7])}
            if mwg:
                res["workgroups"], res["iterations"] = int(mwg.group(1)), int(mwg.group(2))
                res["loop_us_at_2p4GHz"] = round(tot * int(mwg.group(2)) / 2400.0, 2)
        else:
            res["error_prof"] = err
    else:
        res["error_prof"] = f"{a.lib} missing: tools/build_variant.sh prof -DPOPCORN_HEAD_PROF"
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
