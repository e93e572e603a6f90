"""Entry-point counterparts of the reference's ``run_train.py`` / ``run_eval.py`` (same flag names and defaults as
arguments/train.py:10-60 and arguments/eval.py:4-26; argparse instead of configargparse, JSONL instead of wandb).

Order of operations of the training loop is the reference's (run_train.py:146-269):
  to_device -> normalise (+ augment) -> limit1/2/3 gating -> forward(train, padding=False, sparse=True) -> get_loss ->
  x lam_weak -> backward -> clip_grad_norm_(gradient_clip) -> Adam step -> zero_grad; StepLR per epoch; checkpoint dict
  {'model','epoch','iter','optimizer','scheduler'} in ``<save_dir>/experiment_*/last_model.pth`` (:445-456).
"""
from __future__ import annotations

import argparse
import collections
import json
import os
import random
import time

import numpy as np
import torch

from . import ops
from .data import stats
from .data.collate import Population_Dataset_collate_fn
from .data.dataset import SyntheticTestRaster, SyntheticWeaksupDataset
from .data.feed import RegionFeed
from .distributed import FlatReducer, init_from_env
from .model import get_model_kwargs, model_dict
from .utils.transform import default_train_transform


def train_parser():
    p = argparse.ArgumentParser(description="Training Population Estimation (MI355X)")
    p.add_argument("-r", "--resume", type=str)
    p.add_argument("-treg", "--target_regions", nargs="+", default=["pri2017"])
    p.add_argument("-tregtrain", "--target_regions_train", nargs="+", default=["pri2017"])
    p.add_argument("-S1", "--Sentinel1", action="store_true")
    p.add_argument("-S2", "--Sentinel2", action="store_true")
    p.add_argument("-NIR", "--NIR", action="store_true")
    p.add_argument("-wb", "--weak_batch_size", type=int, default=2)
    p.add_argument("-wvb", "--weak_val_batch_size", type=int, default=1)
    p.add_argument("-pret", "--pretrained", action="store_true")
    p.add_argument("-m", "--model", type=str, default="POPCORN")
    p.add_argument("-binit", "--biasinit", type=float, default=0.75)
    p.add_argument("-occmodel", "--occupancymodel", action="store_true")
    p.add_argument("-binp", "--buildinginput", action="store_true")
    p.add_argument("-sinp", "--segmentationinput", action="store_true")
    p.add_argument("-senbuilds", "--sentinelbuildings", action="store_true")
    p.add_argument("-fe", "--feature_extractor", type=str, default="DDA")
    p.add_argument("-e", "--num_epochs", type=int, default=100)
    p.add_argument("-lr", "--learning_rate", type=float, default=1e-4)
    p.add_argument("-l", "--loss", nargs="+", default=["log_l1_loss"])
    p.add_argument("-sreg", "--scale_regularization", default=0.01, type=float)
    p.add_argument("-la", "--lam", nargs="+", type=float, default=[1.0])
    p.add_argument("-lw", "--lam_weak", type=float, default=100.0)
    p.add_argument("-lim1", "--limit1", type=int, default=9000000)
    p.add_argument("-lim2", "--limit2", type=int, default=9000000)
    p.add_argument("-lim3", "--limit3", type=int, default=13000000)
    p.add_argument("-wd", "--weightdecay", type=float, default=0.0)
    p.add_argument("-lrs", "--lr_step", type=int, default=5)
    p.add_argument("-lrg", "--lr_gamma", type=float, default=0.75)
    p.add_argument("-gc", "--gradient_clip", type=float, default=0.01)
    p.add_argument("--save_dir", default="outputs")
    p.add_argument("-w", "--num_workers", type=int, default=0)
    p.add_argument("-lt", "--logstep_train", type=int, default=25)
    p.add_argument("-val", "--val_every_n_epochs", type=int, default=2)
    p.add_argument("-wv", "--weak_validation", action="store_true")
    p.add_argument("-testi", "--test_every_i_steps", type=int, default=500000)
    p.add_argument("-vi", "--val_every_i_steps", type=int, default=500000)
    p.add_argument("--seed", type=int, default=1600)
    p.add_argument("--save-model", default="both", choices=["last", "best", "no", "both"])
    p.add_argument("-mws", "--max_weak_samples", type=int, default=None)
    # build-specific
    p.add_argument("--synthetic_regions", type=int, default=64, help="size of the synthetic weaksup dataset")
    p.add_argument("--synthetic_hw_range", type=int, nargs=2, default=[64, 144],
                   help="side range (pixels) of the synthetic census-region crops; the reference's regions reach ~3000 px sides (limit1 = 9e6 px per batch)")
    p.add_argument("--fixed_hw", type=int, nargs=2, default=None, help="fixed crop size (enables HIP-graph replay)")
    p.add_argument("--torch_optimizer", action="store_true",
                   help="reference recipe through torch autograd + torch.optim.Adam instead of the fused HIP step")
    p.add_argument("--max_steps", type=int, default=None)
    p.add_argument("--synthetic_val_regions", type=int, default=16, help="size of the synthetic weak-validation set")
    p.add_argument("--test_raster_hw", type=int, nargs=2, default=[384, 448], help="synthetic target-test raster (test_target)")
    p.add_argument("--test_patchsize", type=int, default=256)
    p.add_argument("--test_overlap", type=int, default=32)
    p.add_argument("--precision", choices=["fp32", "bf16"], default="fp32",
                   help="bf16: mixed precision (bf16 MFMA operands + channels-last bf16 activations, fp32 master weights; DESIGN.md 7)")
    return p


def eval_parser():
    p = argparse.ArgumentParser(description="Ensemble evaluation (MI355X)")
    p.add_argument("-r", "--resume", nargs="+", type=str, default=[])
    p.add_argument("-treg", "--target_regions", nargs="+", default=["pri2017"])
    p.add_argument("-S1", "--Sentinel1", action="store_true")
    p.add_argument("-S2", "--Sentinel2", action="store_true")
    p.add_argument("-NIR", "--NIR", action="store_true")
    p.add_argument("-m", "--model", type=str, default="POPCORN")
    p.add_argument("-binit", "--biasinit", type=float, default=0.75)
    p.add_argument("-occmodel", "--occupancymodel", action="store_true")
    p.add_argument("-senbuilds", "--sentinelbuildings", action="store_true")
    p.add_argument("-pret", "--pretrained", action="store_true")
    p.add_argument("-fe", "--feature_extractor", type=str, default="DDA")
    p.add_argument("--fourseasons", action="store_true")
    p.add_argument("--save_dir", default="outputs")
    p.add_argument("--seed", type=int, default=1610)
    p.add_argument("--raster_hw", type=int, nargs=2, default=[2304, 2560])
    p.add_argument("--patchsize", type=int, default=2048)
    p.add_argument("--overlap", type=int, default=128)
    p.add_argument("--ensemble", type=int, default=1, help="members to instantiate when no --resume checkpoints are given")
    p.add_argument("--precision", choices=["fp32", "bf16"], default="fp32", help="arithmetic mode of the kernels (DESIGN.md 7)")
    return p


def seed_all(seed):
    """utils/utils.py:50-59."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)


def new_log(folder, args=None):
    """utils/utils.py:62-73."""
    os.makedirs(folder, exist_ok=True)
    n_exp = len(os.listdir(folder))
    randn = round((time.time() * 1000000) % 1000)
    exp = os.path.join(folder, f"experiment_{n_exp}_{randn}")
    os.makedirs(exp, exist_ok=True)
    if args is not None:
        with open(os.path.join(exp, "args.csv"), "w") as fh:
            fh.write("key,value\n")
            for k, v in vars(args).items():
                fh.write(f"{k},{v}\n")
    return exp


def normalize_sample(sample, device, transform=None):
    """to_cuda_inplace + apply_transformations_and_normalize (utils/utils.py:22-43,130-214) on the device: S2 augmentations
    on the raw digital numbers, per-band normalisation + concat [S2, S1] (one HIP kernel), then the joint geometric
    transform of input and admin_mask."""
    s2, s1 = sample["S2"].to(device, non_blocking=True).float(), sample["S1"].to(device, non_blocking=True).float()
    if transform is not None and "S2" in transform:
        s2 = transform["S2"](s2)
    raw = torch.cat([s2, s1], 1).contiguous()
    x = ops.select_normalize(raw, (0, 1, 2, 3, 4, 5), stats.MEAN6, stats.STD6)
    admin = sample["admin_mask"].to(device).float()
    if transform is not None and "general" in transform:
        x, m = transform["general"]((x, admin.unsqueeze(1)))
        admin = m[:, 0]
    out = {"input": x.contiguous(), "admin_mask": admin.contiguous(),
           "census_idx": sample["census_idx"].to(device).contiguous(), "y": sample["y"].to(device).float().contiguous()}
    return out


def prepare_sample_fused(sample, transform=None):
    """The fast form of ``normalize_sample`` for the fused step (round 6): ``sample`` holds DEVICE tensors (``data.feed.RegionFeed``); the
    augmentation parameters are drawn on the host with the reference's generator consumption (``draw_fused_params``), ONE launch
    (``ops.augment_raw``) applies them while it assembles the raw [S2 | S1] tile, and the normalisation happens inside the step's ingest
    (the executor's ``raw`` input form) -- no ``.float()`` / ``torch.cat`` / normalise / flip / rot90 launches, no synchronous copy.
    Returns None when ``transform`` is not the reference trainer's set (the caller then takes ``normalize_sample``)."""
    from .utils.transform import draw_fused_params
    s2, s1, admin = sample["S2"], sample["S1"], sample["admin_mask"]
    if not (s2.is_cuda and s2.dtype == torch.float32 and s1.dtype == torch.float32 and s2.shape[1] == 4 and s1.shape[1] == 2):
        return None
    params = draw_fused_params(transform)
    if params is None:
        return None
    raw, adm = ops.augment_raw(s2.contiguous(), s1.contiguous(), admin.float().contiguous(), params)
    return {"raw": raw, "admin_mask": adm, "census_idx": sample["census_idx"].contiguous(), "y": sample["y"].float().contiguous()}


def limit_regime(num_pix, limit1, limit2, limit3):
    """The memory-driven truncation of the backward pass by batch size in pixels (run_train.py:191-198; defaults
    arguments/train.py:34-36: 9e6 / 9e6 / 13e6): (encoder_no_grad, unet_no_grad, skip the batch).  The limits nest like the
    reference's ifs: limit2 / limit3 only apply beyond limit1 / limit2."""
    enc_ng = unet_ng = skip = False
    if num_pix > limit1:
        enc_ng = True
        if num_pix > limit2:
            unet_ng = True
            if num_pix > limit3:
                skip = True
    return enc_ng, unet_ng, skip


class Trainer:
    def __init__(self, args):
        self.args = args
        self.rank, self.local_rank, self.world = init_from_env()
        if not torch.cuda.is_available():
            raise SystemExit("popcorn_amd training needs a HIP device (no CPU path)")
        torch.cuda.set_device(self.local_rank)
        self.device = torch.device("cuda", self.local_rank)
        self.exp = new_log(os.path.join(args.save_dir, "So2Sat"), args) if self.rank == 0 else None
        # host-side glue (collate, augmentation draws) is many tiny CPU ops: an OpenMP pool as wide as a 256-core host
        # costs more per op than the op itself
        torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
        if self.world > 1 and args.torch_optimizer:
            raise SystemExit("--torch_optimizer is the single-process reference recipe: it has no gradient all-reduce; "
                             "data-parallel runs use the fused step (drop the flag)")
        seed_all(args.seed)
        ds = SyntheticWeaksupDataset(args.synthetic_regions, min_hw=args.synthetic_hw_range[0], max_hw=args.synthetic_hw_range[1],
                                     seed=args.seed, fixed_hw=args.fixed_hw)
        self.sampler = None
        if self.world > 1:
            self.sampler = torch.utils.data.distributed.DistributedSampler(ds, num_replicas=self.world, rank=self.rank,
                                                                           shuffle=True, seed=args.seed, drop_last=True)
        self.loader = torch.utils.data.DataLoader(ds, batch_size=args.weak_batch_size, num_workers=args.num_workers,
                                                  shuffle=self.sampler is None, sampler=self.sampler,
                                                  collate_fn=Population_Dataset_collate_fn, drop_last=True,
                                                  persistent_workers=args.num_workers > 0)
        # weak validation set (run_train.py:410-414: a second Population_Dataset in weaksup mode, batch size -wvb)
        self.val_loader = torch.utils.data.DataLoader(
            SyntheticWeaksupDataset(args.synthetic_val_regions, min_hw=args.synthetic_hw_range[0], max_hw=args.synthetic_hw_range[1],
                                    seed=args.seed + 77, fixed_hw=args.fixed_hw),
            batch_size=args.weak_val_batch_size, shuffle=False, collate_fn=Population_Dataset_collate_fn, drop_last=False)
        self._test_raster = None
        self.model = model_dict[args.model](**get_model_kwargs(args, args.model)).to(self.device)   # same seed: same init on every rank
        self.model.set_precision(args.precision)
        # from here on the CPU generators feed per-rank randomness (augmentation coins, the 60x60 sparsity grid of
        # get_sparsity_mask): every rank gets its own stream, rank 0 keeps the single-process one
        seed_all(args.seed + 2 + 1000 * self.rank)
        self.data_transform = default_train_transform()                    # run_train.py:386-402
        self.reducer = FlatReducer()
        self.info = {"epoch": 0, "iter": 0, "sampleitr": 0}
        self._r2buf = collections.deque()
        if args.torch_optimizer:
            head_name = ["head.6.weight", "head.6.bias"]                     # run_train.py:82-90
            named = list(self.model.named_parameters())
            self.optimizer = torch.optim.Adam([
                {"params": [p for n, p in named if n not in head_name and "unetmodel" not in n], "weight_decay": args.weightdecay},
                {"params": [p for n, p in named if n not in head_name and "unetmodel" in n], "weight_decay": args.weightdecay},
                {"params": [p for n, p in named if n in head_name and "unetmodel" not in n], "weight_decay": 0.0}],
                lr=args.learning_rate)
            self.scheduler = torch.optim.lr_scheduler.StepLR(self.optimizer, step_size=args.lr_step, gamma=args.lr_gamma)
            self.fused = None
        else:
            from .train import FusedTrainStep
            # raw_norm: the feed hands the step the loader's own 6 bands [S2 R G B NIR | S1 VV VH] un-normalised (prepare_sample_fused);
            # band j of that tile IS model channel j
            self.fused = FusedTrainStep(self.model, lr=args.learning_rate, weight_decay=args.weightdecay,
                                        gradient_clip=args.gradient_clip, loss=args.loss, lam=args.lam,
                                        scale_regularization=args.scale_regularization, lam_weak=args.lam_weak,
                                        reducer=self.reducer, use_graph=args.fixed_hw is not None,
                                        raw_norm=(tuple(range(6)), stats.MEAN6, stats.STD6))
        if args.resume:
            self.resume(args.resume)

    # ---- checkpoint (run_train.py:445-476) --------------------------------------------------------------------------
    def save_model(self, prefix="last"):
        if self.rank != 0:
            return None
        path = os.path.join(self.exp, f"{prefix}_model.pth")
        if self.fused is not None:
            # torch.optim.Adam / StepLR state-dict layout (what the reference's resume() loads, run_train.py:458-472),
            # plus the flat buffers themselves under an extra key
            a = self.args
            opt = self.fused.torch_adam_state_dict([n for n, _ in self.model.named_parameters()])
            opt["fused_adam"] = self.fused.optimizer_state()
            ep = self.info["epoch"] - 1                       # StepLR.last_epoch at the time the reference saves
            sched = {"step_size": a.lr_step, "gamma": a.lr_gamma, "base_lrs": [a.learning_rate] * 3, "last_epoch": max(ep, 0),
                     "_step_count": max(ep, 0) + 1, "_get_lr_called_within_step": False, "_last_lr": [self.fused.lr] * 3,
                     "lr": self.fused.lr}
        else:
            opt, sched = self.optimizer.state_dict(), self.scheduler.state_dict()
        torch.save({"model": self.model.state_dict(), "epoch": self.info["epoch"], "iter": self.info["iter"],
                    "optimizer": opt, "scheduler": sched}, path)
        return path

    def _lr_at(self, epoch):
        """StepLR(step_size=lr_step, gamma=lr_gamma) stepped once per finished epoch (run_train.py:93,139-141)."""
        a = self.args
        return a.learning_rate * (a.lr_gamma ** (epoch // a.lr_step)) if a.lr_gamma != 1.0 else a.learning_rate

    def resume(self, path):
        ck = torch.load(path, map_location="cpu", weights_only=False)
        self.model.load_state_dict(ck["model"])
        if self.fused is not None:
            self.fused.sync_from_model()
        self.info["epoch"], self.info["iter"] = ck["epoch"], ck["iter"]
        opt = ck.get("optimizer") or {}
        if self.fused is not None:
            if "fused_adam" in opt:
                self.fused.load_optimizer_state(opt["fused_adam"])
            elif "state" in opt and "param_groups" in opt:
                # a torch.optim.Adam state dict (reference checkpoints, --torch_optimizer runs): per-parameter
                # exp_avg / exp_avg_sq / step -> the flat buffers, matched through the parameter order of run_train.py:82-90
                self.fused.load_torch_adam_state(opt, [n for n, _ in self.model.named_parameters()])
            else:
                print("warning: checkpoint holds no usable optimizer state; Adam moments start from zero")
            self.fused.set_lr(self._lr_at(self.info["epoch"]))       # the checkpoint is written BEFORE the epoch's lr update
        else:
            if "state" in opt and "param_groups" in opt:
                self.optimizer.load_state_dict({"state": opt["state"], "param_groups": opt["param_groups"]})
                self.scheduler.load_state_dict({k: v for k, v in ck["scheduler"].items() if k != "lr"})
            else:
                print("warning: checkpoint holds no torch.optim.Adam state; moments start from zero")

    # ---- loop ------------------------------------------------------------------------------------------------------
    def train_step(self, sample):
        a = self.args
        s = None
        if self.fused is not None and torch.is_tensor(sample.get("S2")) and sample["S2"].is_cuda:
            # a batch staged by the feed: augmentation + assembly in one launch, normalisation inside the step (raw input form)
            s = prepare_sample_fused(sample, self.data_transform)
        if s is None:
            s = normalize_sample(sample, self.device, self.data_transform)
        dk = "raw" if "raw" in s else "input"
        num_pix = s[dk].shape[0] * s[dk].shape[2] * s[dk].shape[3]
        if self.world > 1 and a.fixed_hw is None:
            # variable tile sizes: ranks must take the same regime (and skip together), or their collective counts
            # diverge and the job hangs.  The largest rank decides.
            import torch.distributed as dist
            t = torch.tensor([num_pix], device=self.device, dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            num_pix = int(t.item())
        enc_ng, unet_ng, skip = limit_regime(num_pix, a.limit1, a.limit2, a.limit3)
        if skip:
            return None
        if self.fused is not None:
            # loss_out is one device buffer overwritten by every step: keep a copy for the running log mean
            loss = self.fused.step(s, encoder_no_grad=enc_ng, unet_no_grad=unet_ng)[0].clone()
            # the R2 buffers hold the last 300 samples and are read at log steps only: a step whose samples will have left the window
            # by the next log step does not copy them (two launches per step on the smallest regions)
            to_log = a.logstep_train - (self.info["iter"] % a.logstep_train)
            if to_log * s["y"].numel() <= 300 + s["y"].numel():
                self._buffer_r2(self.fused.last["popcount"], s["y"])
            return loss
        from torch.nn.utils import clip_grad_norm_
        from .utils.losses import get_loss
        out = self.model(s, train=True, padding=False, sparse=True, encoder_no_grad=enc_ng, unet_no_grad=unet_ng)
        loss, _ = get_loss(out, s, scale=out["scale"], loss=a.loss, lam=a.lam, scale_regularization=a.scale_regularization,
                           tag="weak")
        if torch.isnan(loss) or torch.isinf(loss):
            raise Exception("detected NaN/Inf loss..")
        self.optimizer.zero_grad()
        (loss * a.lam_weak).backward()
        if a.gradient_clip > 0.0:
            clip_grad_norm_(self.model.parameters(), a.gradient_clip)
        self.optimizer.step()
        self._buffer_r2(out["popcount"].detach(), s["y"])
        return loss.detach()

    def _buffer_r2(self, pred, y):
        """The 300-sample prediction / target buffers behind the logged training R2 (run_train.py:107-108,220-221,272);
        kept on the device, read at log steps only."""
        self._r2buf.append((pred.detach().clone(), y.detach().clone()))
        while sum(p.numel() for p, _ in self._r2buf) > 300 and len(self._r2buf) > 1:
            self._r2buf.popleft()

    # ---- validation / in-training target test (run_train.py:289-370) ---------------------------------------------------
    def validate_weak(self):
        """Weak validation: census-level metrics of model(sample, padding=False) over the validation regions."""
        from .utils.metrics import get_test_metrics
        self.model.eval()
        pred, gt = [], []
        with torch.no_grad():
            for sample in self.val_loader:
                s = normalize_sample(sample, self.device, None)
                out = self.model(s, padding=False)
                pred.append(out["popcount"])
                gt.append(s["y"])
        stats = get_test_metrics(torch.cat(pred), torch.cat(gt).float(), tag="MainCensus_synthetic_fine")
        self.valweak_stats = {k + "/val": float(v) for k, v in stats.items()}
        self._log(self.valweak_stats)
        return self.valweak_stats

    def test_target(self, save=False):
        """In-training target test: sliding windows over the test raster -> stitched map -> census aggregation -> metrics
        (run_train.py:314-370).  The reference accumulates in fp16 host maps; here the accumulators are fp32 and stay on
        the device (popcorn_amd.eval.Stitcher)."""
        from . import eval as E
        from .utils.metrics import get_test_metrics
        a = self.args
        if self._test_raster is None:
            self._test_raster = SyntheticTestRaster(a.test_raster_hw[0], a.test_raster_hw[1], seasons=1, n_regions=64,
                                                    seed=a.seed + 10, device=self.device)
        data = self._test_raster
        self.model.eval()
        out, _, scale, _ = E.evaluate_raster([self.model], data.raster, a.test_patchsize, a.test_overlap, False,
                                             FlatReducer(), 0) if self.world == 1 else \
            E.evaluate_raster([self.model], data.raster, a.test_patchsize, a.test_overlap, False, self.reducer, self.rank)
        cp, cg = E.convert_popmap_to_census(out, data.boundary, data.census_idx, data.census_pop)
        stats = get_test_metrics(cp, cg, tag="MainCensus_synthetic_fine")
        self.target_test_stats = {k + "/targettest": float(v) for k, v in stats.items()}
        if save and self.rank == 0:
            torch.save({"popdensemap": out.cpu(), "scale": None if scale is None else scale.cpu()},
                       os.path.join(self.exp, "synthetic_predictions.pt"))
        self._log(self.target_test_stats)
        return self.target_test_stats

    def _log(self, record):
        if self.rank == 0:
            with open(os.path.join(self.exp, "train_log.jsonl"), "a") as fh:
                fh.write(json.dumps({**record, **self.info}) + "\n")

    def train(self):
        a = self.args
        log = open(os.path.join(self.exp, "train_log.jsonl"), "a") if self.rank == 0 else None
        lr = self._lr_at(self.info["epoch"])
        t0 = time.time()
        recent = collections.deque(maxlen=a.logstep_train)                  # running window for the log line, across epochs
        losses = []
        for epoch in range(self.info["epoch"], a.num_epochs):
            self.model.train()
            if self.sampler is not None:
                self.sampler.set_epoch(epoch)                              # a fresh shuffle per epoch on every rank
            losses = []
            # the fused step is fed one batch ahead through pinned memory and a copy stream (data/feed.py: RegionFeed); the torch-optimizer
            # recipe keeps the reference's synchronous loop
            feed = RegionFeed(self.loader, self.device) if self.fused is not None else self.loader
            for i, sample in enumerate(feed):
                loss = self.train_step(sample)
                if loss is not None:
                    losses.append(loss)
                    recent.append(loss)
                self.info["iter"] += 1
                self.info["sampleitr"] += a.weak_batch_size
                if (i + 1) % a.val_every_i_steps == 0 and a.weak_validation:          # run_train.py:255-259
                    self.validate_weak()
                    self.model.train()
                if (i + 1) % a.test_every_i_steps == 0:                              # run_train.py:262-265
                    self.test_target(save=True)
                    self.model.train()
                if self.info["iter"] % a.logstep_train == 0 and recent:              # by global iteration (run_train.py:240)
                    window = torch.stack(list(recent))
                    # the fused step never reads the loss back; its NaN/Inf guard (run_train.py:224-227) is this one
                    # device->host read per log step, covering every step of the window
                    finite = torch.isfinite(window).all()
                    if self.fused is not None:
                        # the conv / hidden-layer ReLU epilogues take the IEEE maximum (a NaN activation becomes 0), so a
                        # diverged run shows up in the gradient norm and the parameters before it shows up in the loss
                        finite = finite & torch.isfinite(self.fused.norm).all() & torch.isfinite(self.fused.flat_p.sum())
                    if not bool(finite):
                        raise Exception("detected NaN/Inf loss..")
                    if log:
                        from .utils.losses import r2
                        pr = torch.cat([p_ for p_, _ in self._r2buf])[-300:]
                        yy = torch.cat([y_ for _, y_ in self._r2buf])[-300:]
                        rec = {"iter": self.info["iter"], "epoch": epoch, "loss": window.mean().item(), "lr": lr,
                               "Population_weak/r2": float(r2(pr, yy)) if pr.numel() > 1 else 0.0}
                        log.write(json.dumps(rec) + "\n")
                        log.flush()
                if a.max_steps and self.info["iter"] >= a.max_steps:
                    break
            if losses and not bool(torch.isfinite(torch.stack(losses)).all()):
                raise Exception("detected NaN/Inf loss..")
            self.info["epoch"] = epoch + 1
            if a.save_model in ("last", "both"):
                self.save_model("last")
            if (epoch + 1) % a.val_every_n_epochs == 0:                              # run_train.py:126-137
                if a.weak_validation:
                    self.validate_weak()
                self.test_target(save=True)
                if a.save_model in ("last", "both"):
                    self.save_model("last")
            if a.lr_gamma != 1.0:                                          # StepLR(step_size=lr_step, gamma), run_train.py:93,141
                lr = self._lr_at(epoch + 1)
                if self.fused is not None:
                    self.fused.set_lr(lr)
                else:
                    self.scheduler.step()
            if a.max_steps and self.info["iter"] >= a.max_steps:
                break
        if self.rank == 0:
            print(f"Training finished in {time.time() - t0:.1f} s, {self.info['iter']} iterations; logs in {self.exp}")
        return losses


def run_train(argv=None):
    args = train_parser().parse_args(argv)
    t = Trainer(args)
    t.train()
    return t


def run_eval(argv=None):
    from . import eval as E
    from .utils.metrics import get_test_metrics
    args = eval_parser().parse_args(argv)
    rank, local_rank, world = init_from_env()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    seed_all(args.seed)
    models = []
    n = max(len(args.resume), args.ensemble)
    for j in range(n):
        m = model_dict[args.model](**get_model_kwargs(args, args.model)).to(dev)
        if j < len(args.resume):
            m.load_state_dict(torch.load(args.resume[j], map_location="cpu", weights_only=False)["model"])   # run_eval.py:243-257
        models.append(m.eval().set_precision(args.precision))
    data = SyntheticTestRaster(args.raster_hw[0], args.raster_hw[1], seasons=4 if args.fourseasons else 1, device=dev)
    reducer = FlatReducer()
    t0 = time.time()
    out, out_std, scale, scale_std = E.evaluate_raster(models, data.raster, args.patchsize, args.overlap, args.fourseasons,
                                                       reducer, rank)
    res = {}
    cp, cg = E.convert_popmap_to_census(out, data.boundary, data.census_idx, data.census_pop)
    res.update({k: float(v) for k, v in get_test_metrics(cp, cg, tag="MainCensus_synthetic_fine").items()})
    adj = E.adjust_map_to_census(out.clone(), data.boundary, data.census_idx, data.census_pop)
    cp, cg = E.convert_popmap_to_census(adj, data.boundary, data.census_idx, data.census_pop)
    res.update({k: float(v) for k, v in get_test_metrics(cp, cg, tag="AdjCensus_synthetic_fine").items()})
    torch.cuda.synchronize()
    if rank == 0:
        res["seconds"] = time.time() - t0
        print(json.dumps(res))
    return res
