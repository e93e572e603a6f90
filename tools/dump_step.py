"""One eager fp32 (or bf16) train step on the seeded bench batch; dumps loss + every parameter gradient.  A/B of two library builds:
    POPCORN_HIP_LIB=ab/libpopcorn_base.so python tools/dump_step.py /tmp/a.pt; python tools/dump_step.py /tmp/b.pt
    python tools/dump_step.py --cmp /tmp/a.pt /tmp/b.pt        (bit-level comparison, tensor by tensor)"""
import sys
import torch
sys.path.insert(0, ".")


def main():
    if sys.argv[1] == "--cmp":
        a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
        nd = 0
        for k in a:
            if not torch.equal(a[k], b[k]):
                d = (a[k].double() - b[k].double()).abs().max().item()
                s = a[k].double().abs().max().item()
                print("DIFF", k, "max abs", d, "rel", d / max(s, 1e-30))
                nd += 1
        print(nd, "of", len(a), "tensors differ")
        return
    prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
    from popcorn_amd.model import POPCORN
    from popcorn_amd.train import FusedTrainStep
    from popcorn_amd import ops
    from popcorn_amd.data import stats
    from popcorn_amd.data.synthetic import make_raw_batch
    torch.manual_seed(1600)
    m = POPCORN(input_channels=6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
    m.set_precision(prec)
    tr = FusedTrainStep(m, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01, use_graph=False)
    batch = make_raw_batch(64, 100, 100, seed=1600, device="cuda")
    smp = {"input": ops.select_normalize(batch["raw"], stats.BAND6, stats.MEAN6, stats.STD6), "admin_mask": batch["admin_mask"],
           "census_idx": batch["census_idx"], "y": batch["y"]}
    torch.manual_seed(7)
    loss = tr.step(smp)
    torch.cuda.synchronize()
    d = {"loss": loss.detach().float().cpu().clone()}
    for n, g in tr.grads.items():
        d[n] = g.detach().float().cpu().clone()
    torch.save(d, sys.argv[1])
    print("saved", len(d), "tensors; loss", d["loss"].tolist())


main()
