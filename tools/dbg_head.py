import sys, os
sys.path.insert(0, os.getcwd())
import torch, torch.nn.functional as F
from oracle import popcorn_oracle as O
from popcorn_amd import ops
G = "tests/golden"
def _mk(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed); return torch.randn(*shape, generator=g) * scale
B, H, W, Hp, Wp, py, px = 1, 37, 29, 64, 64, 13, 17
sd = O.load_golden_state(G)
names = [f"head.{i}.{n}" for i in (0, 2, 4, 6) for n in ("weight", "bias")]
for route in ["pc", "pd", "sm", "const"]:
    work = dict(sd)
    for n in names: work[n] = sd[n].clone().requires_grad_(True)
    feat = _mk(B, 16, Hp, Wp, seed=21).requires_grad_(True)
    gen = torch.Generator().manual_seed(22)
    building = torch.rand(B, 1, H, W, generator=gen)
    admin = (torch.rand(B, H, W, generator=gen) < 0.6).float() * 5.0
    census = torch.full((B,), 5, dtype=torch.int64)
    g_pc = torch.randn(B, generator=gen); g_pd = torch.randn(B, H, W, generator=gen) * 0.1; g_sm = torch.randn(B, H, W, generator=gen) * 0.1
    headin = feat[:, :, py:py + H, px:px + W]
    out = O.head_forward(work, headin)[:, 0]
    scale = F.relu(out); pd = scale * building[:, 0]; pc = (pd * (admin == census.view(-1, 1, 1))).sum((1, 2))
    loss = {"pc": (pc * g_pc).sum(), "pd": (pd * g_pd).sum(), "sm": (scale * g_sm).sum(), "const": 0.37 * scale.sum()}[route]
    loss.backward()
    ht = [sd[n].cuda() for n in names]
    kw = dict(g_popcount=g_pc.cuda() if route == "pc" else None, g_popdense=g_pd.cuda() if route == "pd" else None,
              g_scale_map=g_sm.cuda() if route == "sm" else None, g_scale_const=torch.tensor([0.37], device="cuda") if route == "const" else None)
    grads, g_feat = ops.head_bwd(feat.detach().cuda(), py, px, H, W, ht, building.cuda(), admin_mask=admin.cuda(), census_idx=census.cuda(), **kw)
    print("route", route)
    for n, gr in zip(names, grads):
        ref = work[n].grad
        print("  %-14s maxabs ref %.4e  err %.4e" % (n, ref.abs().max().item(), (gr.cpu() - ref).abs().max().item()))
    print("  g_feat        maxabs ref %.4e  err %.4e" % (feat.grad.abs().max().item(), (g_feat.cpu() - feat.grad).abs().max().item()))
