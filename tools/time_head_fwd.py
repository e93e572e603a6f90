import sys, torch
sys.path.insert(0, ".")
from popcorn_amd import ops, _lib as L
from popcorn_amd.model import POPCORN
prec = sys.argv[1]
torch.manual_seed(0)
m = POPCORN(input_channels=6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
B, H, W = 64, 100, 100
with L.precision(prec):
    feat = L.as_act(torch.randn(B, 16, 128, 128, device="cuda"))
    bld = torch.rand(B, 1, H, W, device="cuda")
    adm = torch.ones(B, H, W, device="cuda"); cen = torch.ones(B, dtype=torch.int64, device="cuda")
    ht = m.head_tensors()
    for _ in range(3): ops.head_fwd(feat, 14, 14, H, W, ht, bld, admin_mask=adm, census_idx=cen)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.head_fwd(feat, 14, 14, H, W, ht, bld, admin_mask=adm, census_idx=cen)
    e1.record(); torch.cuda.synchronize()
    print(prec, "head_fwd call", e0.elapsed_time(e1) / 20 * 1e3, "us")
