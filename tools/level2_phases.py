"""Phase timeline of the one-launch 32 x 32 level backward (pc_level2_bwd_group): wall-clock stamps of workgroup (0, 0), B = 64,
2 problems.   python3 tools/level2_phases.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.getcwd())
import torch
from popcorn_amd import ops, _lib as L

B = 64
BF = "--bf16" in sys.argv
if BF:
    L.lib().pc_set_precision(L.PC_PREC_BF16)
ts = torch.zeros(16, dtype=torch.int64, device="cuda")
wb = ops.WgradBatch(torch.device("cuda"))
probs = []
for i in range(2):
    z = lambda *s: torch.randn(*s, device="cuda")
    if BF:
        za = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    else:
        za = z
    one = torch.ones(16, device="cuda")
    probs.append({"g2": za(B, 16, 32, 32), "c1": torch.relu(za(B, 16, 32, 32)), "x": torch.relu(za(B, 16, 32, 32)), "w1": z(16, 16, 3, 3) * .1,
                  "w2": z(16, 16, 3, 3) * .1, "bn1": L.bn(None, one, one * 0, one * 0, one), "act": torch.relu(za(B, 16, 64, 64)),
                  "act_bn": L.bn(None, one, one * 0, one * 0, one), "out": za(B, 16, 64, 64), "dw1": z(16, 16, 3, 3), "db1": z(16),
                  "dw2": z(16, 16, 3, 3), "db2": z(16), "_k": one})
names_bf = ["loads + staging", "dgrad -> G1 (registers)", "wgrad dW2", "reduce dW2", "restage x | G1", "dgrad G1", "wgrad dW1", "reduce dW1",
            "pool scatter"]
names = names_bf if BF else ["stage G2, c1 + weights", "wgrad dW2", "barrier", "dgrad -> G1", "stage x (+ w1)", "wgrad dW1", "dgrad -> Gp", "pool scatter"]
for it in range(3):
    L.lib().pc_debug_level2_ts(C.c_void_p(ts.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    wb.level2_bwd_group(probs)
    e1.record()
    wb.finish()
    torch.cuda.synchronize()
    t = ts.cpu().tolist()
    print(f"launch {e0.elapsed_time(e1) * 1e3:.1f} us; workgroup (0,0): " +
          ", ".join(f"{n} {(t[i + 1] - t[i]) / 100:.1f}" for i, n in enumerate(names)) + f"; total {(t[len(names)] - t[0]) / 100:.1f} us")
L.lib().pc_debug_level2_ts(None)
