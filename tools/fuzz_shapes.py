"""Random input geometries through the drop-in forward vs the CPU oracle (ad-hoc fuzz; the committed tests pin a few)."""
import os, sys, random
sys.path.insert(0, os.getcwd())
import torch
from oracle import popcorn_oracle as O
from popcorn_amd.model import POPCORN
torch.manual_seed(1600)
m = POPCORN(input_channels=6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda().eval()
sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = 0.0
special = [(1, 100, 100), (2, 128, 128), (1, 64, 96), (1, 36, 68), (1, 98, 226), (3, 50, 50)]
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 24):
    B, H, W = special[it] if it < len(special) else (rnd.randint(1, 3), rnd.randint(29, 150), rnd.randint(29, 150))
    padding = rnd.random() < 0.5
    g = torch.Generator().manual_seed(it)
    x = torch.randn(B, 6, H, W, generator=g)
    with torch.no_grad():
        ref = O.popcorn_forward(sd, {"input": x}, padding=padding)
        out = m({"input": x.cuda()}, padding=padding)
    e1 = ((out["popdensemap"].cpu() - ref["popdensemap"]).abs().max() / ref["popdensemap"].abs().max()).item()
    e2 = ((out["popcount"].cpu() - ref["popcount"]).abs().max() / ref["popcount"].abs().max()).item()
    worst = max(worst, e1, e2)
    print(f"B={B} H={H} W={W} padding={padding}: rel err map {e1:.2e} count {e2:.2e}", flush=True)
    assert e1 < 1e-4 and e2 < 1e-4
print("worst", worst)
