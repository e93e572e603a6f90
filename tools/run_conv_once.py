"""Run the 8->8 @128x128 B=64 forward conv a few times (for rocprofv3 counter collection)."""
import ctypes as C, sys, os
sys.path.insert(0, os.getcwd())
import torch
from popcorn_amd import ops, _lib as L
lib = L.lib()
B, cin, cout, hw = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 8, 8, 128
x = torch.randn(B, cin, hw, hw, device="cuda"); w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.1
b = torch.zeros(cout, device="cuda"); out = torch.empty(B, cout, hw, hw, device="cuda")
bnd = L.bn(b); sa, d = L.src(x), L.dst(out)
for _ in range(6):
    lib.pc_conv3x3_bn_relu_fwd(C.byref(sa), None, L.ptr(w), C.byref(bnd), 1, C.byref(d), B, hw, hw, cin, cout, L.stream_ptr())
torch.cuda.synchronize()
