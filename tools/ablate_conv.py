"""Ablation of the conv3x3 kernel (8->8 @128x128, B=64): which phase costs what.  GPU only."""
import ctypes as C, sys, os
sys.path.insert(0, os.getcwd())
import torch
from popcorn_amd import ops, _lib as L
lib = L.lib()
B = 64
for (cin, cout, hw) in [(8, 8, 128), (16, 8, 128)]:
    x = torch.randn(B, cin, hw, hw, device="cuda"); w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.1
    b = torch.zeros(cout, device="cuda"); out = torch.empty(B, cout, hw, hw, device="cuda")
    bnd = L.bn(b)
    sa, d = L.src(x), L.dst(out)
    st = L.stream_ptr()
    def run(n):
        for _ in range(n):
            lib.pc_conv3x3_bn_relu_fwd(C.byref(sa), None, L.ptr(w), C.byref(bnd), 1, C.byref(d), B, hw, hw, cin, cout, st)
    for dbg, grid, tag in [(0, 0, "full"), (1, 0, "no loader"), (2, 0, "no mfma"), (4, 0, "no store"), (3, 0, "no loader+mfma"), (7, 0, "nothing"), (8, 0, "empty kernel"), (5, 0, "mfma only"), (6, 0, "loader only"), (0, 1024, "full grid1024"), (0, 384, "full grid384"),
                           (0, 512, "full grid512"), (0, 2048, "full grid2048"), (0, 256, "full grid256")]:
        lib.pc_debug_conv(dbg, grid)
        run(5); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            st = L.stream_ptr()
            with torch.cuda.graph(g, stream=s):
                st = L.stream_ptr(); run(20)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        st = L.stream_ptr()
        print(f"{cin}->{cout}@{hw}  {tag:16s} {e0.elapsed_time(e1) * 1e3 / 20:8.1f} us")
    lib.pc_debug_conv(0, 0)
