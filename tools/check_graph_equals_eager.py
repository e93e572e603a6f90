import os, sys
sys.path.insert(0, os.getcwd())
import torch
from popcorn_amd import ops
from popcorn_amd.data import stats
from popcorn_amd.data.synthetic import make_raw_batch
from popcorn_amd.model import Args, get_model_kwargs, model_dict
from popcorn_amd.train import FusedTrainStep
dev = torch.device("cuda")
margs = Args(Sentinel1=True, NIR=True, Sentinel2=True, feature_extractor="DDA", occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True)
for use_graph in (False, True):
    torch.manual_seed(1600)
    model = model_dict["POPCORN"](**get_model_kwargs(margs, "POPCORN")).to(dev)
    tr = FusedTrainStep(model, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01, loss=("log_l1_loss",), lam=(1.0,), scale_regularization=0.01, lam_weak=100.0, use_graph=use_graph)
    batch = make_raw_batch(64, 100, 100, seed=1600, device=dev)
    sample = tr.static_buffers(64, 100, 100)
    sample["admin_mask"].copy_(batch["admin_mask"]); sample["census_idx"].copy_(batch["census_idx"]); sample["y"].copy_(batch["y"])
    torch.manual_seed(1600)
    out = []
    for it in range(12):
        ops.select_normalize(batch["raw"], stats.BAND6, stats.MEAN6, stats.STD6, out=sample["input"])
        l = tr.step(sample)
        torch.cuda.synchronize()
        out.append(round(l[0].item(), 6))
    print("graph" if use_graph else "eager", out, "step_count", tr.step_count.tolist())
