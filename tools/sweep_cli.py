"""Robustness sweep of the two entry points (run_train.py / run_eval.py counterparts, popcorn_amd/cli.py) over flag combinations: modality
flags, -occmodel / -senbuilds, loss lists, optimizer path, precision, fixed / variable crop sizes, seasons, ensemble size.  One short epoch /
one small raster each; exceptions and non-finite results are the target."""
import itertools
import os
import sys
import tempfile
import traceback

sys.path.insert(0, os.getcwd())
import torch                                              # noqa: E402
from popcorn_amd import cli                               # noqa: E402

bad = n = 0
tmp = tempfile.mkdtemp()
MOD = ["-S2 -NIR -S1", "-S1", "-S2 -NIR"]
for mod, occ, senb, loss, opt, prec, fixed in itertools.product(MOD, ("-occmodel", ""), ("-senbuilds", ""),
                                                               ("-l log_l1_loss", "-l l1_loss mse_loss -la 1.0 0.001"), ("", "--torch_optimizer"),
                                                               ("fp32", "bf16"), ("--fixed_hw 64 64", "")):
    if opt and prec == "bf16" and False:
        continue
    n += 1
    argv = (f"{mod} {occ} {senb} -pret -wd 1e-5 --biasinit 0.9407 -lr 1e-4 {loss} {opt} --precision {prec} {fixed} --synthetic_regions 6 -wb 2 "
            f"--save_dir {tmp} -lt 100 -val 100 -e 1 --save-model no").split()
    try:
        t = cli.Trainer(cli.train_parser().parse_args(argv))
        t.train()
        torch.cuda.synchronize()
        ps = torch.cat([p.detach().reshape(-1).float() for p in t.model.parameters()])
        if not bool(torch.isfinite(ps).all()):
            bad += 1
            print("TRAIN", " ".join(argv[:-12]), "non-finite parameters BAD", flush=True)
    except Exception as e:
        bad += 1
        print("TRAIN", " ".join(argv), f"EXCEPTION {type(e).__name__}: {str(e)[:300]}", flush=True)
        traceback.print_exc(limit=6)
for mod, occ, senb, four, ens, prec in itertools.product(MOD, ("-occmodel", ""), ("-senbuilds", ""), ("--fourseasons", ""), (1, 2), ("fp32", "bf16")):
    n += 1
    argv = f"{mod} {occ} {senb} -pret --biasinit 0.9407 {four} --ensemble {ens} --precision {prec} --raster_hw 300 420 --patchsize 128 --overlap 16 --save_dir {tmp}".split()
    try:
        res = cli.run_eval(argv)
        if not all(v == v and abs(v) != float("inf") for v in res.values() if isinstance(v, float)):
            bad += 1
            print("EVAL", " ".join(argv), "non-finite metric BAD", res, flush=True)
    except Exception as e:
        bad += 1
        print("EVAL", " ".join(argv), f"EXCEPTION {type(e).__name__}: {str(e)[:300]}", flush=True)
        traceback.print_exc(limit=6)
print(f"{n} invocations, bad: {bad}")
