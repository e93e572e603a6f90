"""Do independent grouped conv launches overlap their ramp/drain when issued on two streams?  GPU only."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from popcorn_amd import ops, _lib as L
B, NSETS, REPS = 64, 4, 10
for (cin, cout, hw) in [(8, 8, 128), (16, 16, 64), (16, 16, 32)]:
    sets = []
    for s in range(NSETS):
        probs = []
        for i in range(4):
            a = torch.randn(B, cin, hw, hw, device="cuda")
            w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.1
            bias = torch.zeros(cout, device="cuda")
            probs.append({"a": a, "w": w, "bn": L.bn(bias), "out": torch.empty(B, cout, hw, hw, device="cuda"), "_k": bias})
        sets.append(probs)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for nstream in (1, 2):
        g = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream()
        for s in sets:
            ops.conv3x3_fwd_group(s)
        torch.cuda.synchronize()
        with torch.cuda.stream(cap):
            with torch.cuda.graph(g, stream=cap):
                if nstream == 2:
                    for st in streams:
                        st.wait_stream(cap)
                for r in range(REPS):
                    for i, s in enumerate(sets):
                        if nstream == 1:
                            ops.conv3x3_fwd_group(s)
                        else:
                            with torch.cuda.stream(streams[i % 2]):
                                ops.conv3x3_fwd_group(s)
                if nstream == 2:
                    for st in streams:
                        cap.wait_stream(st)
        torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        print(f"{cin}->{cout}@{hw} x4  streams={nstream}  {e0.elapsed_time(e1) * 1e3 / (REPS * NSETS):7.1f} us per launch", flush=True)
