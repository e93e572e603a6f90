"""Would decoupled roles overlap?  An MFMA-only conv launch (no loads, no stores) on one stream next to a
loads+stores-only launch of the same kernel on another stream, both at half the resident grid, versus each alone and
versus the normal kernel.  GPU only."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from popcorn_amd import ops, _lib as L
lib = L.lib()
B, REPS = 64, 10
cin, cout, hw = 8, 8, 128
def mk():
    probs = []
    for i in range(4):
        a = torch.randn(B, cin, hw, hw, device="cuda")
        w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.1
        bias = torch.zeros(cout, device="cuda")
        probs.append({"a": a, "w": w, "bn": L.bn(bias), "out": torch.empty(B, cout, hw, hw, device="cuda"), "_k": bias})
    return probs
sets = [mk() for _ in range(8)]
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

def run(plan, tag):
    """plan: list of (dbg, grid, stream index or None)"""
    lib.pc_debug_conv(0, 0)
    for s in sets:
        ops.conv3x3_fwd_group(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); cap = torch.cuda.Stream()
    with torch.cuda.stream(cap):
        with torch.cuda.graph(g, stream=cap):
            sa.wait_stream(cap); sb.wait_stream(cap)
            for r in range(REPS):
                for i in range(0, 8, 2):
                    for j, (dbg, grid, st) in enumerate(plan):
                        lib.pc_debug_conv(dbg, grid)
                        if st is None:
                            ops.conv3x3_fwd_group(sets[i + j])
                        else:
                            with torch.cuda.stream((sa, sb)[st]):
                                ops.conv3x3_fwd_group(sets[i + j])
            cap.wait_stream(sa); cap.wait_stream(sb)
    lib.pc_debug_conv(0, 0)
    torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"{tag:60s} {e0.elapsed_time(e1) * 1e3 / (REPS * 4):7.1f} us per step of the plan", flush=True)

run([(0, 0, None), (0, 0, None)], "2 x full kernel, back to back")
run([(5, 0, None), (5, 0, None)], "2 x MFMA-only, back to back (full grid)")
run([(2, 0, None), (2, 0, None)], "2 x loads+stores-only, back to back (full grid)")
run([(5, 128, None), (5, 128, None)], "2 x MFMA-only, back to back (half grid)")
run([(2, 128, None), (2, 128, None)], "2 x loads+stores-only, back to back (half grid)")
run([(5, 128, 0), (2, 128, 1)], "MFMA-only || loads+stores-only (two streams, half grids)")
run([(5, 128, 0), (5, 128, 1)], "MFMA-only || MFMA-only (two streams, half grids)")
run([(2, 128, 0), (2, 128, 1)], "loads+stores || loads+stores (two streams, half grids)")
run([(0, 128, 0), (0, 128, 1)], "full || full (two streams, half grids)")
