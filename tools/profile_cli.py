"""Where does the host time of the CLI training loop go?  (cProfile over 100 steps; GPU box)"""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.getcwd())
from popcorn_amd.cli import Trainer, train_parser
argv = "-S2 -NIR -S1 -occmodel -senbuilds -pret -wd 1e-5 --biasinit 0.9407 --synthetic_regions 256 -wb 32 --fixed_hw 96 96 --num_epochs 100 --max_steps 120 -lt 1000 --save_dir /tmp/prof_cli".split()
try:
    t = Trainer(train_parser().parse_args(argv))
except SystemExit:
    argv = argv[:-2]
    t = Trainer(train_parser().parse_args(argv))
t.args.max_steps = 20
t.train()                 # warm-up incl. graph capture
t.args.max_steps = 120
pr = cProfile.Profile(); pr.enable()
t.train()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
