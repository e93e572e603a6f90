"""Single-node data parallelism: one process per GPU, torch.distributed (backend 'nccl' = RCCL over xGMI).

The reference has no distributed code; this is the build's own design for BASELINE configs 4-5:

  * tiles are independent samples -> each rank draws its own B_local tiles; no data-path collective in the forward;
  * the only couplings are the batch mean of the loss and the mean over *all selected pixels* of the scale
    regulariser (utils/losses.py:53,74).  Each rank therefore normalises by the GLOBAL batch size and the GLOBAL
    Nsel, which needs one 16-byte all-reduce of {Nsel, sum(scale)} before the backward pass;
  * gradients: ONE sum-all-reduce of the flat 39,298-element fp32 gradient buffer (157 KB) per step -- never
    per-tensor.  At this size a ring over 8 GPUs is latency-bound (~275 KB per GPU, ~2 us of xGMI time), so one
    collective on a pre-flattened buffer is the right shape; bucketing would only add launches.
    Because every rank already used the global normalisers, SUM (not mean) reproduces the single-process gradient;
  * clip_grad_norm_ and Adam then run identically on every rank on the reduced buffer.

The same code runs on the gloo backend with CPU tensors (tests/test_distributed_cpu.py).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def force_collectives():
    """POPCORN_DIST_FORCE=1: take the multi-rank code path (process group, split graphs, both all-reduces) also with ONE rank --
    the RCCL calls of the data-parallel step on a single-GPU box (tests/test_gpu_dp.py)."""
    return os.environ.get("POPCORN_DIST_FORCE") == "1"


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* (as set by torch.distributed.run).
    Returns (rank, local_rank, world).  No-op for world == 1 (unless POPCORN_DIST_FORCE=1)."""
    rank, local_rank, world = env_world()
    if (world > 1 or force_collectives()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # POPCORN_DIST_BACKEND=gloo: functional runs of the multi-rank paths on a box with fewer GPUs than ranks
            backend = os.environ.get("POPCORN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


class FlatReducer:
    """Sum-all-reduce of the flat gradient buffer and of the 2-scalar loss statistics."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (dist.is_initialized() and force_collectives())      # collectives are issued
        self.capture_failed = False
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        if self.active and self.backend == "nccl" and torch.cuda.is_available():
            # communicator set-up (and RCCL's lazy channel allocation) happens on the first collective: do it here, on every
            # rank alike, so that the first captured step records plain collective nodes
            t = torch.zeros(2, device="cuda", dtype=torch.float64)
            dist.all_reduce(t, group=group)
            t32 = torch.zeros(64, device="cuda", dtype=torch.float32)
            dist.all_reduce(t32, group=group)
            torch.cuda.synchronize()

    def capturable(self):
        """Collectives of this group can be recorded into a HIP graph (torch's NCCL / RCCL backend supports stream capture;
        gloo works through the host and cannot).  OPT-IN (POPCORN_DP_ONE_GRAPH=1): the one-graph step has only ever run with one
        rank (single-GPU test boxes), where the collectives are identities; until a run with >= 2 GPUs has verified it the
        default is the split form -- three graphs with the two collectives launched eagerly between them (+ ~30 us per step)."""
        return self.active and self.backend == "nccl" and not self.capture_failed and os.environ.get("POPCORN_DP_ONE_GRAPH", "0") == "1"

    def all_agree(self, ok: bool) -> bool:
        """True iff ``ok`` holds on EVERY rank (eager MIN all-reduce of a flag): decisions that change the sequence of collectives
        a rank issues must be taken by all ranks together."""
        if not self.active:
            return bool(ok)
        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return bool(t.item())

    def _host_path_sync(self, t: torch.Tensor):
        # gloo moves device tensors through the host and waits for the producing stream itself; with two ranks sharing one GPU
        # (the functional runs) that wait stalls for 30 - 1,700 ms every few steps (tools/dp_gloo_probe.py: 2-rank step 270 ms
        # instead of 4 ms).  Draining the stream first -- the collective is a host round trip anyway -- removes the stalls.
        if self.backend != "nccl" and t.is_cuda:
            torch.cuda.current_stream(t.device).synchronize()

    def reduce_stats(self, stats: torch.Tensor):
        """stats: float64[2] {Nsel, sum(scale)} -> global sums (in place)."""
        if self.active:
            self._host_path_sync(stats)
            dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=self.group)
        return stats

    def reduce_grads(self, flat: torch.Tensor):
        """flat fp32 gradient buffer -> sum over ranks (in place).  One collective per step."""
        if self.active:
            self._host_path_sync(flat)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        return flat

    def global_batch(self, local_batch: int) -> int:
        return local_batch * self.world


def shard_indices(n_items: int, rank: int, world: int):
    """Round-robin assignment of independent work items (tiles / sliding windows) to ranks."""
    return list(range(rank, n_items, world))
