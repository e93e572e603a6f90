"""bf16 mixed precision (PC_PREC_BF16): how far is the HIP path from the oracle's bf16 restatement, and how far are both
from fp32?  Prints the error table the tolerances of tests/test_gpu_bf16.py are taken from."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import popcorn_oracle as O  # noqa: E402
from popcorn_amd.model import POPCORN  # noqa: E402
from popcorn_amd.train import FusedTrainStep  # noqa: E402


def rel(a, b):
    return ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


def main():
    g = np.load("tests/golden/g5_train.npz")
    sample = {k: torch.from_numpy(g[k]) for k in ("input", "admin_mask", "census_idx", "y")}
    for seed in (1600, 1601):
        torch.manual_seed(seed)
        m = POPCORN(input_channels=6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
        sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        for padding in (True, False):
            with torch.no_grad():
                o32 = O.popcorn_forward(sd, {"input": sample["input"].clone()}, padding=padding)
                with O.bf16_mode():
                    o16 = O.popcorn_forward(sd, {"input": sample["input"].clone()}, padding=padding)
                m.set_precision("bf16")
                h16 = m({"input": sample["input"].cuda()}, padding=padding)
                m.set_precision("fp32")
                h32 = m({"input": sample["input"].cuda()}, padding=padding)
            for k in ("popdensemap", "popcount"):
                print(f"seed {seed} pad {int(padding)} {k:12s} hip16-vs-o16 {rel(h16[k].cpu(), o16[k]):.2e}   o16-vs-o32 {rel(o16[k], o32[k]):.2e}"
                      f"   hip16-vs-o32 {rel(h16[k].cpu(), o32[k]):.2e}   hip32-vs-o32 {rel(h32[k].cpu(), o32[k]):.2e}")
        # train step
        torch.manual_seed(3)
        l32, out32, g32, _ = O.train_step_grads(sd, dict(sample))
        with O.bf16_mode():
            torch.manual_seed(3)
            l16, out16, g16, _ = O.train_step_grads(sd, dict(sample))
        m.set_precision("bf16")
        tr = FusedTrainStep(m, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01)
        torch.manual_seed(3)
        loss = tr.step({k: v.cuda() for k, v in sample.items()})
        torch.cuda.synchronize()
        print(f"seed {seed} loss hip16 {loss[0].item():.6f} o16 {l16.item():.6f} o32 {l32.item():.6f}")
        print(f"   popcount hip16-vs-o16 {rel(tr.last['popcount'].cpu(), out16['popcount']):.2e}  o16-vs-o32 {rel(out16['popcount'], out32['popcount']):.2e}")
        e_h = {n: rel(tr.grads[n].cpu(), g16[n]) for n in g16}
        e_o = {n: rel(g16[n], g32[n]) for n in g16}
        worst = max(e_h, key=e_h.get)
        print(f"   grads hip16-vs-o16: worst {e_h[worst]:.2e} ({worst}), median {np.median(list(e_h.values())):.2e};"
              f"   o16-vs-o32: worst {max(e_o.values()):.2e}, median {np.median(list(e_o.values())):.2e}")
        for n in list(g16)[:6] + list(g16)[-8:]:
            print(f"      {n:60s} {e_h[n]:.2e}   (o16-vs-o32 {e_o[n]:.2e})")


if __name__ == "__main__":
    main()
