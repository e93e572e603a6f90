"""Fixed cost of one launch of each kernel family: a tiny problem (B=1, 32x32) replayed back to back in a HIP graph."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from popcorn_amd import ops, _lib as L
dev = "cuda"
def timeit(fn, tag, n=50):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); cap = torch.cuda.Stream()
    with torch.cuda.stream(cap):
        with torch.cuda.graph(g, stream=cap):
            for _ in range(n):
                fn()
    torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"{tag:44s} {e0.elapsed_time(e1) * 1e3 / n:6.2f} us per launch", flush=True)
B, H, W = 1, 32, 32
for cin, cout in [(8, 8), (16, 16), (32, 8)]:
    x = torch.randn(B, cin, H, W, device=dev); w = torch.randn(cout, cin, 3, 3, device=dev); b = torch.zeros(cout, device=dev)
    bn = L.bn(b); out = torch.empty(B, cout, H, W, device=dev)
    probs = [{"a": x, "w": w, "bn": bn, "out": out} for _ in range(4)]
    timeit(lambda: ops.conv3x3_fwd_group(probs), f"conv fwd {cin}->{cout} x4 problems")
    g = torch.randn(B, cout, H, W, device=dev); gi = torch.empty(B, cin if cin <= 16 else 16, H, W, device=dev)
    act = torch.randn_like(gi)
    cn = gi.shape[1]
    dprobs = [{"g": g, "w": w, "out": gi, "act": act, "act_bn": L.bn(None)} for _ in range(2)]
    timeit(lambda: ops.conv3x3_dgrad_group(dprobs, 0, cn), f"conv dgrad {cout}->{cn} x2 problems")
    wb = ops.WgradBatch(torch.device(dev))
    dw = torch.empty(cout, cin, 3, 3, device=dev); db = torch.empty(cout, device=dev)
    def wg():
        wb2 = ops.WgradBatch(torch.device(dev))
        wb2.conv3x3_group([{"a": x, "g": g, "dw": dw, "db": db} for _ in range(2)], cout)
        return wb2
    timeit(lambda: wg(), f"conv wgrad {cin}->{cout} x2 problems (stage 1)")
z = torch.zeros(64, device=dev)
timeit(lambda: z.add_(1.0), "torch elementwise on 64 floats")
