"""Round 6: the fused conv backward (pc_conv3x3_bwd_group, fp32 mode) at the step's geometries (B = 64 tiles, two streams), in both
multiplication forms (pc_set_conv_split 1 / 0), next to the data-gradient + weight-gradient launches the 16-channel forms replace.
Every timed call runs on rotating buffer sets (cold Infinity Cache), N launches captured into one HIP graph, HIP events around three replays.

    python3 tools/time_conv_bwd.py [--iters 20] [--json gpurun_out/conv_bwd.json]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from popcorn_amd import ops, _lib as L  # noqa: E402


def timed(fn, iters):
    """us per launch: `iters` launches (rotating buffer sets) captured into ONE HIP graph and replayed (the Python side of a grouped call,
    ~40 us of descriptor marshalling, would otherwise bound every launch shorter than that)"""
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
    ap.add_argument("--sets", type=int, default=3)
    ap.add_argument("--json", default=None)
    ap.add_argument("--ablate", action="store_true", help="the 8<->8 @128x128 launch with phases of the split kernel switched off (needs a -DPOPCORN_CONV_ABLATE build: tools/build_variant.sh ablate -DPOPCORN_CONV_ABLATE; POPCORN_HIP_LIB=ab/libpopcorn_ablate.so)")
    a = ap.parse_args()
    dev = torch.device("cuda")
    B = a.batch
    res = []
    bn8 = lambda: L.bn(None, torch.rand(8, device=dev) + 0.5, torch.zeros(8, device=dev), torch.zeros(8, device=dev), torch.rand(8, device=dev) + 0.5)  # noqa: E731

    def mk(*shape):
        return torch.randn(*shape, device=dev)

    def run(label, build, forms=(1, 0)):
        for form in forms:
            prev = L.lib().pc_set_conv_split(form)
            try:
                sets = [build() for _ in range(a.sets)]
                fns = [s for s in sets]

                def call(i=0):
                    fns[i % len(fns)]()
                us = timed(call, a.iters)
            except L.PopcornHipError as e:
                us = None
                print(label, "form", form, "->", e)
            finally:
                L.lib().pc_set_conv_split(prev)
            res.append({"launch": label, "form": "split" if form else "fp32mfma", "us": us})
            print(json.dumps(res[-1]), flush=True)

    # 8 <-> 8 @128 x 128, two streams (inc2, up1b, up1a skip block)
    def b88(H, W, n):
        def build():
            g = [mk(B, 8, H, W) for _ in range(n)]
            x = [torch.relu(mk(B, 8, H, W)) for _ in range(n)]
            w = [mk(8, 8, 3, 3) * 0.2 for _ in range(n)]
            out = [torch.empty(B, 8, H, W, device=dev) for _ in range(n)]
            dw = [torch.empty(8, 8, 3, 3, device=dev) for _ in range(n)]
            db = [torch.empty(8, device=dev) for _ in range(n)]
            bns = [bn8() for _ in range(n)]

            def f():
                wb = ops.WgradBatch(dev)
                wb.conv3x3_bwd_group([{"g": g[i], "x": x[i], "w": w[i], "out": out[i], "dw": dw[i], "db": db[i], "x_bn": bns[i]}
                                      for i in range(n)], 8, 0)
            return f
        return build
    if a.ablate:
        for dbg, what in [(0, "full"), (1, "no split / LDS writes"), (2, "no dgrad matrix phase"), (4, "no wgrad matrix phase"), (6, "no matrix phases"),
                          (8, "no prefetch loads"), (16, "no epilogue"), (24, "no loads, no epilogue"), (7, "loads + epilogue only"),
                          (31, "loop skeleton"), (25, "matrix phases only"), (9, "no loads, no split")]:
            L.lib().pc_debug_conv_bwd(dbg)
            run(f"8<->8 @128x128 x2, dbg {dbg}: {what}", b88(128, 128, 2), forms=(1,))
        L.lib().pc_debug_conv_bwd(0)
        return
    run("dgrad+wgrad 8<->8 @128x128 x2", b88(128, 128, 2))
    run("dgrad+wgrad 8<->8 @64x64 x2", b88(64, 64, 2))

    # 16 <-> 16 @64 x 64 (down1 second conv): two column blocks per stream, one launch (split form only)
    def b1616():
        g = [mk(B, 16, 64, 64) for _ in range(2)]
        x = [torch.relu(mk(B, 16, 64, 64)) for _ in range(2)]
        w = [mk(16, 16, 3, 3) * 0.2 for _ in range(2)]
        out = [torch.empty(B, 16, 64, 64, device=dev) for _ in range(2)]
        dw = [torch.empty(16, 16, 3, 3, device=dev) for _ in range(2)]
        db = [torch.empty(16, device=dev) for _ in range(2)]
        bns = [[bn8() for _ in range(2)] for _ in range(2)]

        def f():
            wb = ops.WgradBatch(dev)
            wb.conv3x3_bwd_group([{"g": g[s], "x": x[s][:, 8 * i:8 * i + 8], "w": w[s], "out": out[s][:, 8 * i:8 * i + 8], "dw": dw[s],
                                   "db": db[s] if i == 0 else None, "c0_add": 8 * i, "x_bn": bns[s][i]} for s in range(2) for i in range(2)],
                                 16, 0)
        return f
    run("dgrad+wgrad 16<->16 @64x64 x2 (two column blocks each)", b1616, forms=(1,))

    def b1616_pair():
        g = [mk(B, 16, 64, 64) for _ in range(2)]
        x = [torch.relu(mk(B, 16, 64, 64)) for _ in range(2)]
        w = [mk(16, 16, 3, 3) * 0.2 for _ in range(2)]
        out = [torch.empty(B, 16, 64, 64, device=dev) for _ in range(2)]
        dw = [torch.empty(16, 16, 3, 3, device=dev) for _ in range(2)]
        db = [torch.empty(16, device=dev) for _ in range(2)]
        bn16 = [L.bn(None, torch.rand(16, device=dev) + 0.5, torch.zeros(16, device=dev), torch.zeros(16, device=dev),
                     torch.rand(16, device=dev) + 0.5) for _ in range(2)]

        def f():
            wb = ops.WgradBatch(dev)
            wb.conv3x3_group([{"a": x[s], "g": g[s], "dw": dw[s], "db": db[s]} for s in range(2)], 16)
            ops.conv3x3_dgrad_group([{"g": g[s], "w": w[s], "out": out[s], "act": x[s], "act_bn": bn16[s]} for s in range(2)], 0, 16)
        return f
    run("wgrad 16->16 + dgrad 16->16 @64x64 x2 (the pair it replaces)", b1616_pair, forms=(0,))

    # Down block first conv (d1a): g 16 @64x64, x = pooled a2 (8 @64x64), scatter into 8 @128x128
    def bpool():
        g = [mk(B, 16, 64, 64) for _ in range(2)]
        act = [torch.relu(mk(B, 8, 128, 128)) for _ in range(2)]
        x = [torch.nn.functional.max_pool2d(t, 2) for t in act]
        w = [mk(16, 8, 3, 3) * 0.2 for _ in range(2)]
        out = [torch.zeros(B, 8, 128, 128, device=dev) for _ in range(2)]
        dw = [torch.empty(16, 8, 3, 3, device=dev) for _ in range(2)]
        db = [torch.empty(16, device=dev) for _ in range(2)]
        bns = [bn8() for _ in range(2)]

        def f():
            wb = ops.WgradBatch(dev)
            wb.conv3x3_bwd_group([{"g": g[s], "x": x[s], "w": w[s], "out": out[s], "dw": dw[s], "db": db[s], "x_bn": bns[s],
                                   "pool_act": act[s]} for s in range(2)], 8, 0, accumulate=True)
        return f
    run("dgrad(pool scatter)+wgrad 16<->8 @64x64 x2", bpool, forms=(1,))

    def bpool_pair():
        g = [mk(B, 16, 64, 64) for _ in range(2)]
        act = [torch.relu(mk(B, 8, 128, 128)) for _ in range(2)]
        x = [torch.nn.functional.max_pool2d(t, 2) for t in act]
        w = [mk(16, 8, 3, 3) * 0.2 for _ in range(2)]
        out = [torch.zeros(B, 8, 128, 128, device=dev) for _ in range(2)]
        dw = [torch.empty(16, 8, 3, 3, device=dev) for _ in range(2)]
        db = [torch.empty(16, device=dev) for _ in range(2)]
        bns = [bn8() for _ in range(2)]

        def f():
            wb = ops.WgradBatch(dev)
            wb.conv3x3_group([{"a": x[s], "g": g[s], "dw": dw[s], "db": db[s]} for s in range(2)], 16)
            ops.conv3x3_dgrad_group([{"g": g[s], "w": w[s], "out": out[s], "act": act[s], "act_bn": bns[s]} for s in range(2)], 0, 8,
                                    pool=True, accumulate=True)
        return f
    run("wgrad 8->16 + dgrad 16->8 pool @64x64 x2 (the pair it replaces)", bpool_pair, forms=(0,))

    if a.json:
        os.makedirs(os.path.dirname(a.json) or ".", exist_ok=True)
        with open(a.json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
