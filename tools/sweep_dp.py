"""Data-parallel sweep on ONE GPU (two gloo ranks sharing the device): regime x precision x graph x unequal selections -- two ranks on the
halves of a batch must reproduce the single-process parameters (the global normalisers {B, Nsel} make SUM-all-reduce exact)."""
import itertools
import os
import socket
import sys

sys.path.insert(0, os.getcwd())
import torch                                              # noqa: E402
import torch.multiprocessing as mp                        # noqa: E402

REG = {0: {}, 1: dict(encoder_no_grad=True), 2: dict(encoder_no_grad=True, unet_no_grad=True)}


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def run(rank, world, port, prec, graph, reg, q):
    import torch.distributed as dist
    from oracle import popcorn_oracle as O
    from popcorn_amd.data.synthetic import make_raw_batch
    from popcorn_amd.distributed import FlatReducer
    from popcorn_amd.model import POPCORN
    from popcorn_amd.train import FusedTrainStep
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    torch.manual_seed(1600)
    m = POPCORN(input_channels=6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
    m.set_precision(prec)
    tr = FusedTrainStep(m, lr=1e-3, weight_decay=1e-5, gradient_clip=0.01, reducer=FlatReducer(), use_graph=graph)
    b = make_raw_batch(4, 100, 100, seed=3, region="disc")          # disc radii differ per sample: unequal Nsel per rank
    full = {"input": O.select_normalize(b["raw"]), "admin_mask": b["admin_mask"], "census_idx": b["census_idx"], "y": b["y"]}
    idx = list(range(rank, 4, world))
    s = {k: v[idx].cuda() for k, v in full.items()}
    for step in range(3):
        torch.manual_seed(100 + step)
        tr.step(s, **REG[reg])
    torch.cuda.synchronize()
    if rank == 0:
        q.put(tr.flat_p.cpu().numpy().tolist())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def launch(world, prec, graph, reg):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=run, args=(r, world, port, prec, graph, reg, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return torch.tensor(out)


if __name__ == "__main__":
    bad = 0
    for prec, graph, reg in itertools.product(("fp32", "bf16"), (False, True), (0, 1, 2)):
        p1, p2 = launch(1, prec, graph, reg), launch(2, prec, graph, reg)
        e = ((p1 - p2).abs().max() / p1.abs().max()).item()
        tol = 2e-5 if prec == "fp32" else 2e-3
        ok = e <= tol
        bad += 0 if ok else 1
        print(f"{prec} graph={int(graph)} regime={reg}: 2-rank vs 1-rank parameter distance {e:.2e} {'ok' if ok else 'BAD'}", flush=True)
    print("bad:", bad)
