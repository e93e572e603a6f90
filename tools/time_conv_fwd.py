"""Round 6: the forward conv launches of the step (B = 64 tiles, 4 problems = 2 networks x 2 streams) in the split-operand form
(csrc/conv3x3_fwd_s3.h; pc_set_conv_split 1, or 2 = also the plain 8 -> 8 layers) next to the fp32-MFMA form (0).  Every timed call runs on
rotating buffer sets (cold Infinity Cache), N launches captured into one HIP graph, HIP events around three replays.

    python3 tools/time_conv_fwd.py [--iters 20] [--json gpurun_out/conv_fwd.json]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from popcorn_amd import ops, _lib as L  # noqa: E402


def timed(fn, iters):
    """us per launch: `iters` launches (rotating buffer sets) captured into ONE HIP graph and replayed -- the Python side of a grouped call
    (descriptor marshalling, ~40 us) would otherwise bound every launch shorter than that"""
    fn(0)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for i in range(iters):
                fn(i)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * iters)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--nprob", type=int, default=4)
    ap.add_argument("--sets", type=int, default=3)
    ap.add_argument("--json", default=None)
    ap.add_argument("--ablate", action="store_true", help="two launches with phases of the split kernel switched off (pc_debug_conv)")
    a = ap.parse_args()
    dev = torch.device("cuda")
    B, K = a.batch, a.nprob
    res = []

    def bn(c):
        t = [torch.randn(c, device=dev) * 0.1, torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1,
             torch.randn(c, device=dev) * 0.1, torch.rand(c, device=dev) + 0.5]
        return L.bn(t[0], t[1], t[2], t[3], t[4], 1e-5), t

    def run(label, build, forms, nbytes):
        for form in forms:
            prev = L.lib().pc_set_conv_split(form)
            try:
                fns = [build() for _ in range(a.sets)]

                def call(i=0):
                    fns[i % len(fns)]()
                us = timed(call, a.iters)
            finally:
                L.lib().pc_set_conv_split(prev)
            res.append({"launch": label, "form": {0: "fp32mfma", 1: "split", 2: "split (all shapes)"}[form], "us": us,
                        "alg_MB": nbytes / 1e6, "TBps": nbytes / us / 1e6})
            print(json.dumps(res[-1]), flush=True)

    def plain(ci, co, H, W, pool):
        def build():
            x = [torch.relu(torch.randn(B, ci, H, W, device=dev)) for _ in range(K)]
            w = [torch.randn(co, ci, 3, 3, device=dev) * 0.2 for _ in range(K)]
            out = [torch.empty(B, co, H, W, device=dev) for _ in range(K)]
            bns = [bn(co) for _ in range(K)]
            po = [ops.pool_out_like(o) if pool else None for o in out]
            probs = []
            for i in range(K):
                pr = {"a": x[i], "w": w[i], "bn": bns[i][0], "out": out[i]}
                if pool:
                    pr["pool_out"] = po[i]
                probs.append(pr)

            def f():
                ops.conv3x3_fwd_group(probs)
            f.keep = (x, w, out, bns, po)
            return f
        return build

    def composed(Cs, H, W):
        def build():
            sk = [torch.relu(torch.randn(B, Cs, H, W, device=dev)) for _ in range(K)]
            z = [torch.relu(torch.randn(B, Cs, H // 2, W // 2, device=dev)) for _ in range(K)]
            w = [torch.randn(8, 2 * Cs, 3, 3, device=dev) * 0.1 for _ in range(K)]
            wt = [torch.randn(Cs, Cs, 2, 2, device=dev) * 0.2 for _ in range(K)]
            bt = [torch.randn(Cs, device=dev) for _ in range(K)]
            out = [torch.empty(B, 8, H, W, device=dev) for _ in range(K)]
            bns = [bn(8) for _ in range(K)]
            probs = [{"skip": sk[i], "z": z[i], "w": w[i], "wt": wt[i], "bt": bt[i], "bn": bns[i][0], "out": out[i]} for i in range(K)]
            wss = ops.conv3x3_up_compose([{"w": w[i], "wt": wt[i], "bt": bt[i]} for i in range(K)])
            for pr, ws in zip(probs, wss):
                pr["ws"] = ws

            def f():
                ops.conv3x3_up_fwd_group(probs)
            f.keep = (sk, z, w, wt, bt, out, bns, wss)
            return f
        return build

    px = lambda h, w: B * K * h * w * 4.0  # noqa: E731
    if a.ablate:
        for dbg, what in [(0, "full"), (2, "no matrix phase"), (32, "no split / LDS writes"), (64, "no loads"), (4, "no epilogue"),
                          (2 | 32, "loads + epilogue"), (1 | 4, "matrix phase only"), (1 | 2, "epilogue only"), (2 | 4 | 32, "loads only"),
                          (2 | 4, "loads + split"), (1 | 2 | 4, "loop skeleton")]:
            L.lib().pc_debug_conv(dbg, 0)
            run(f"up1a composed @128x128, dbg {dbg}: {what}", composed(8, 128, 128), (1,), px(128, 128) * 16 + px(64, 64) * 8)
            run(f"d1b 16 -> 16 @64x64 + pool, dbg {dbg}: {what}", plain(16, 16, 64, 64, True), (1,), px(64, 64) * 32 + px(32, 32) * 16)
        L.lib().pc_debug_conv(0, 0)
        if a.json:
            with open(a.json, "w") as f:
                json.dump(res, f, indent=1)
        return
    run("up1a composed 8 + 8z -> 8 @128x128", composed(8, 128, 128), (1, 0), px(128, 128) * 16 + px(64, 64) * 8)
    run("up2a composed 16 + 16z -> 8 @64x64", composed(16, 64, 64), (1, 0), px(64, 64) * 24 + px(32, 32) * 16)
    run("d1b 16 -> 16 @64x64 + pooled output", plain(16, 16, 64, 64, True), (1, 0), px(64, 64) * 32 + px(32, 32) * 16)
    run("d1a 8 -> 16 @64x64", plain(8, 16, 64, 64, False), (1, 0), px(64, 64) * 24)
    run("inc2 8 -> 8 @128x128 + pooled output", plain(8, 8, 128, 128, True), (1, 0), px(128, 128) * 16 + px(64, 64) * 8)
    run("up1b 8 -> 8 @128x128", plain(8, 8, 128, 128, False), (1, 0), px(128, 128) * 16)
    run("up2b 8 -> 8 @64x64", plain(8, 8, 64, 64, False), (1, 0), px(64, 64) * 16)
    if a.json:
        os.makedirs(os.path.dirname(a.json) or ".", exist_ok=True)
        with open(a.json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
