"""Data-parallel training on the device: two ranks (gloo backend, both on the one GPU of the test box -- RCCL needs one
GPU per rank) run FusedTrainStep on the two halves of a batch, eagerly and with the split HIP graphs; the updated
parameters must equal a single-process step on the whole batch."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(B, seed=3):
    from oracle import popcorn_oracle as O
    from popcorn_amd.data.synthetic import make_raw_batch
    b = make_raw_batch(B, 64, 48, seed=seed, region="disc")
    return {"input": O.select_normalize(b["raw"]), "admin_mask": b["admin_mask"], "census_idx": b["census_idx"],
            "y": b["y"]}


def _run(rank, world, port, use_graph, q):
    import torch.distributed as dist
    from popcorn_amd.distributed import FlatReducer
    from popcorn_amd.model import POPCORN
    from popcorn_amd.train import FusedTrainStep
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    rccl1 = world in ("rccl1", "rccl1split")   # ONE rank on the real backend, the multi-rank code path forced (POPCORN_DIST_FORCE)
    if rccl1:
        os.environ.update(POPCORN_DIST_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
                          POPCORN_DP_ONE_GRAPH="0" if world == "rccl1split" else "1")
        split = world == "rccl1split"
        world = 1
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1)
    elif world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    torch.manual_seed(1600)
    m = POPCORN(input_channels=6, occupancymodel=True, pretrained=True, biasinit=0.9407, sentinelbuildings=True).cuda()
    tr = FusedTrainStep(m, lr=1e-3, weight_decay=1e-5, gradient_clip=0.01, reducer=FlatReducer(), use_graph=use_graph)
    full = _make(4)
    idx = list(range(rank, 4, world))
    s = {k: v[idx].cuda() for k, v in full.items()}
    losses = []
    for step in range(3):
        torch.manual_seed(100 + step)            # same selection grid on every rank
        l = tr.step(s)
        losses.append(l[0].item())
    torch.cuda.synchronize()
    if rccl1:
        assert tr.reducer.active and dist.get_backend() == "nccl"
        # the sharded-inference collectives (eval.Stitcher.all_reduce: fp32 planes + the int16 count map through int32) on RCCL
        from popcorn_amd.eval import Stitcher
        st = Stitcher(40, 56, "cuda")
        st.acc.copy_(torch.rand(4, 40, 56))
        st.count.copy_(torch.randint(0, 9, (40, 56), dtype=torch.int16))
        a0, c0 = st.acc.clone(), st.count.clone()
        st.all_reduce(tr.reducer)
        torch.cuda.synchronize()
        assert torch.equal(st.acc, a0) and torch.equal(st.count, c0) and st.count.dtype == torch.int16
        # the band form on RCCL: reduce_scatter_tensor (in place) + finalise + all_gather_into_tensor == finalise of the whole
        ref = Stitcher(40, 56, "cuda")
        ref.acc.copy_(a0); ref.count.copy_(c0)
        want = [t.clone() for t in ref.finalize()]
        assert st.reduce_scatter(tr.reducer, 0) == (0, 40)
        st.finalize()
        got = st.gather_bands(tr.reducer)
        torch.cuda.synchronize()
        for a, b in zip(got, want):
            assert torch.equal(a.nan_to_num(-1.0), b.nan_to_num(-1.0))
        if use_graph:
            # one graph with both collectives captured as nodes (POPCORN_DP_ONE_GRAPH=1, opt-in until verified on >= 2 GPUs), or
            # forward | stats all-reduce | backward | gradient all-reduce | update as three graphs around two eager collectives
            assert len(tr._graphs[3]) == (3 if split else 1), (len(tr._graphs[3]), tr.reducer.capture_failed)
    if rank == 0:
        q.put((tr.flat_p.cpu().numpy().tolist(), losses))    # plain lists: no shared-memory handles that die with the child
    if world > 1 or rccl1:
        dist.barrier()
        dist.destroy_process_group()


def _get(q, procs, timeout=600):
    """q.get that gives up as soon as a child has died without reporting (instead of sitting out the whole timeout)."""
    import queue
    import time
    t0 = time.time()
    while True:
        try:
            return q.get(timeout=2)
        except queue.Empty:
            if any(p.exitcode not in (None, 0) for p in procs):
                raise RuntimeError("a rank died: " + str([p.exitcode for p in procs]))
            if time.time() - t0 > timeout:
                raise


def _launch(world, use_graph):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, world, port, use_graph, q)) for r in range(1 if isinstance(world, str) else world)]
    for p in procs:
        p.start()
    out = _get(q, procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return out


@pytest.mark.parametrize("use_graph,world", [(False, 2), (True, 2), (False, 4)])
def test_n_rank_fused_step_equals_single_process(use_graph, world):
    """2 (and, eager = the native executor's FWD | BWD | UPD phase split around the two collectives, 4) gloo ranks sharing the one GPU of
    the test box against the single-process step on the whole batch."""
    p1, l1 = _launch(1, use_graph)
    p2, l2 = _launch(world, use_graph)
    # rank 0 reports only its local part of the batch-mean loss; parameters are what must agree
    p1, p2 = torch.tensor(p1), torch.tensor(p2)
    scale = p1.abs().max().item()
    assert (p1 - p2).abs().max().item() <= 2e-5 * scale, (p1 - p2).abs().max().item()
    assert all(abs(a) < 1e6 for a in l2)


@pytest.mark.parametrize("use_graph,mode", [(False, "rccl1"), (True, "rccl1"), (True, "rccl1split")])
def test_rccl_collectives_in_the_graph_step(use_graph, mode):
    """The data-parallel step with its REAL backend: one rank on nccl (= RCCL; a second rank would need a second GPU), the
    multi-rank code path forced -- process group, float64 stats all-reduce, flat-gradient all-reduce, both CAPTURED INSIDE the
    one replayed HIP graph (``rccl1``) or issued eagerly between three replayed graphs (``rccl1split``,
    POPCORN_DP_ONE_GRAPH=0).  With one rank the sums are identities: parameters and losses must equal the plain
    single-process step exactly."""
    p1, l1 = _launch(1, use_graph)
    pr, lr = _launch(mode, use_graph)
    assert pr == p1 and lr == l1
