"""bf16 head backward: cooperative kernel against the one-role kernel on the bench batch (B = 64, every pixel selected; the
multi-iteration path of both).  Run twice with POPCORN_HEAD_BWD_COOP=0 / 1 and compare the dumped gradients:
    POPCORN_HEAD_BWD_COOP=0 python tools/ab_head_bwd.py /tmp/g0.pt; POPCORN_HEAD_BWD_COOP=1 python tools/ab_head_bwd.py /tmp/g1.pt
    python tools/ab_head_bwd.py --cmp /tmp/g0.pt /tmp/g1.pt"""
import sys
import torch
sys.path.insert(0, ".")


def main():
    if sys.argv[1] == "--cmp":
        a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
        worst = 0.0
        for k in a:
            d = (a[k].double() - b[k].double()).abs().max().item()
            s = a[k].double().abs().max().item()
            worst = max(worst, d / max(s, 1e-30))
            if d / max(s, 1e-30) > 1e-3:
                print("DIFF", k, d, s)
        print("worst relative difference", worst)
        return
    from popcorn_amd.model import POPCORN
    from popcorn_amd.train import FusedTrainStep
    from popcorn_amd import ops
    from popcorn_amd.data import stats
    from popcorn_amd.data.synthetic import make_raw_batch
    torch.manual_seed(1600)
    m = POPCORN(input_channels=6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
    m.set_precision("bf16")
    tr = FusedTrainStep(m, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01, use_graph=False)
    batch = make_raw_batch(64, 100, 100, seed=1600, device="cuda")
    sample = {"input": ops.select_normalize(batch["raw"], stats.BAND6, stats.MEAN6, stats.STD6), "admin_mask": batch["admin_mask"],
              "census_idx": batch["census_idx"], "y": batch["y"]}
    torch.manual_seed(3)
    loss = tr.step(sample)
    torch.cuda.synchronize()
    out = {k: v.detach().cpu().clone() for k, v in tr.grads.items()}
    out["_loss"] = loss.detach().cpu().clone()
    torch.save(out, sys.argv[1])
    print("loss", loss[0].item())


if __name__ == "__main__":
    main()
