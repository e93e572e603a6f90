"""Ablation of the grouped conv3x3 forward launch as the train step issues it: 4 problems (2 networks x 2 streams),
B=64, distinct buffers, rotated over NSETS buffer sets so that the 256 MiB Infinity Cache is cold.  GPU only.

    python3 tools/ablate_conv_group.py [cin cout hw]
"""
import os
import sys
sys.path.insert(0, os.getcwd())
import torch
from popcorn_amd import ops, _lib as L

lib = L.lib()
PREC = os.environ.get("ABL_PREC", "fp32")          # bf16: the bf16-container instantiations (half the bytes)
lib.pc_set_precision(L.PRECISIONS[PREC])
ADT = L.act_dtype()
ESZ = 2 if PREC == "bf16" else 4
B, NSETS, REPS = 64, 4, 10
cfgs = [(8, 8, 128), (16, 8, 128), (16, 16, 64), (32, 8, 64)]
if len(sys.argv) == 4:
    cfgs = [tuple(int(v) for v in sys.argv[1:4])]
variants = [(0, 0, "full"), (1, 0, "no loader"), (2, 0, "no mfma"), (4, 0, "no store"), (5, 0, "mfma only"),
            (6, 0, "loader only"), (3, 0, "store only"), (7, 0, "nothing"), (0, 128, "full 2 WG/CU"),
            (0, 256, "full 4 WG/CU"), (0, 384, "full 6 WG/CU")]
if os.environ.get("ABL_ONE"):
    variants = [(int(os.environ["ABL_ONE"], 0), 0, "dbg=" + os.environ["ABL_ONE"])]
elif os.environ.get("ABL_ONLY_FULL"):
    variants = [(0, 0, "full"), (16, 0, "old grid"), (0, 0, "full"), (16, 0, "old grid")]
for (cin, cout, hw) in cfgs:
    sets = []
    for s in range(NSETS):
        probs = []
        for i in range(4):
            ca = cin if cin <= 8 else cin // 2
            mk = torch.zeros if os.environ.get("ABL_ZERO") else torch.randn       # zero operands: the power-limit test
            a = L.as_act(mk(B, ca, hw, hw, device="cuda"))
            b = L.as_act(mk(B, cin - ca, hw, hw, device="cuda")) if cin > ca else None
            w = mk(cout, cin, 3, 3, device="cuda") * 0.1
            bias = torch.zeros(cout, device="cuda")
            probs.append({"a": a, "b": b, "w": w, "bn": L.bn(bias), "out": L.empty_act(B, cout, hw, hw, "cuda"),
                          "_keep": bias})
        sets.append(probs)
    flop = 4 * B * hw * hw * 2 * 9 * cin * cout
    byts = 4 * B * hw * hw * ESZ * (cin + cout)
    for dbg, grid, tag in variants:
        lib.pc_debug_conv(dbg, grid)
        for s in sets:
            ops.conv3x3_fwd_group(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()          # graph replay: no CPU launch floor between the kernels
        cap = torch.cuda.Stream()
        with torch.cuda.stream(cap):
            with torch.cuda.graph(g, stream=cap):
                for _ in range(REPS):
                    for s in sets:
                        ops.conv3x3_fwd_group(s)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (REPS * NSETS)
        print(f"{cin:2d}->{cout:2d}@{hw:3d} x4  {tag:14s} {us:7.1f} us   {flop / us / 1e6:6.1f} TFLOP/s  {byts / us / 1e6:5.2f} TB/s",
              flush=True)
        del g
    lib.pc_debug_conv(0, 0)
    del sets
