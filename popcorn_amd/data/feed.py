"""Double-buffered host -> device feed of the fused training step (the reference's loop copies every batch synchronously:
run_train.py:186, utils/utils.py:22-27).  Batch i + 1 moves from PINNED host memory straight into the idle one of the trainer's two static
input sets (``FusedTrainStep.static_buffers(slot=0 / 1)``, one captured graph each) on a copy stream while the graph of the other set
computes step i; two events per set order the two streams.

**The copy stream must not share a hardware queue with the compute stream.**  HIP multiplexes its streams onto a handful of hardware
queues (4 by default) round-robin in creation order, and torch creates its pool of 32 streams at once: every fourth ``torch.cuda.Stream()``
lands on the compute stream's queue.  The SDMA copy itself still runs beside the kernels, but the marker packet behind it (the ``copied``
event) sits in the SHARED in-order queue in front of the next step's kernels, so copy and step serialise: measured on MI355X with B = 64
15-band tiles 2.51 ms per step instead of 1.73 (copy 0.83 ms + step 1.67 ms; ``tools/h2d_legs.py --alias-scan``: streams 3 and 7 of 10).
Which pool stream aliases depends on what the process created before -- round 4's bench legs were "upside-down" (the narrowest feed the
slowest) for exactly this reason.  High-priority streams are no way out (their queues alias too, and an aliased one halves the step
rate).  ``pick_copy_stream`` therefore MEASURES: a pinned copy on the candidate next to a spin kernel on the compute stream either overlaps
(about max of the two) or serialises (about their sum); the first candidate that overlaps is kept."""
from __future__ import annotations

import time

import torch

_PICKED = {}


def pick_copy_stream(device=None, candidates=8, nbytes=16 << 20, verbose=False):
    """A side stream whose copies overlap kernels of the CURRENT stream (see the module docstring).  Cached per (device, current
    stream).  Falls back to the best candidate measured if none overlaps cleanly."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    cur = torch.cuda.current_stream(dev)
    key = (dev.index, cur.cuda_stream)
    if key in _PICKED:
        return _PICKED[key]
    host = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    ev = torch.cuda.Event()

    def copy_alone(st):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        with torch.cuda.stream(st):
            dst.copy_(host, non_blocking=True)
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0

    first = torch.cuda.Stream(dev)
    copy_alone(first)
    t_copy = min(copy_alone(first) for _ in range(3))
    # a spin kernel of about the copy's duration on the compute stream (cycles from a calibration run of torch's own sleep kernel)
    probe_cycles = 1 << 20
    torch.cuda._sleep(probe_cycles)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    torch.cuda._sleep(probe_cycles)
    torch.cuda.synchronize(dev)
    per_cycle = max((time.perf_counter() - t0) / probe_cycles, 1e-10)
    spin = int(t_copy / per_cycle)
    best, seen = None, []
    cand = first
    for k in range(candidates):
        times = []
        for _ in range(3):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            with torch.cuda.stream(cand):
                dst.copy_(host, non_blocking=True)
                ev.record(cand)                      # the marker packet behind the copy: what blocks a shared queue
            torch.cuda._sleep(spin)                  # ... enqueued behind it on the compute stream
            torch.cuda.synchronize(dev)
            times.append(time.perf_counter() - t0)
        t = min(times)
        seen.append((hex(cand.cuda_stream), round(t / t_copy, 2)))
        if best is None or t < best[1]:
            best = (cand, t)
        if t < 1.45 * t_copy:                        # overlapped: about max(copy, spin) = 1x; serialised: about 2x
            best = (cand, t)
            break
        cand = torch.cuda.Stream(dev)
    if verbose:
        print(f"pick_copy_stream: copy {t_copy * 1e3:.3f} ms; (stream, time / copy): {seen} -> {hex(best[0].cuda_stream)}")
    _PICKED[key] = best[0]
    return best[0]


class HostFeed:
    """Double-buffered feed of a ``FusedTrainStep`` (``use_graph=True``) from pinned host batches.

        feed = HostFeed(trainer, B, H, W, kind="raw", raw_channels=6)        # or kind="split" / "input"
        for batch in loader:               # dict of PINNED host tensors keyed like ``feed.sets[0]`` ("raw" + "_packed", ...)
            loss = feed.step(batch)

    ``step(host_batch)`` enqueues the copy of ``host_batch`` into the idle set and runs the step on the set filled by the PREVIOUS call
    (software pipeline of depth one: the first call only copies and returns None; ``flush()`` runs the last batch)."""

    def __init__(self, trainer, B, H, W, kind="raw", raw_channels=6, copy_stream=None):
        self.tr = trainer
        if kind == "split":
            self.sets = [trainer.static_buffers(B, H, W, split=True, slot=s) for s in (0, 1)]
        elif kind == "raw":
            self.sets = [trainer.static_buffers(B, H, W, raw_channels=raw_channels, slot=s) for s in (0, 1)]
        else:
            self.sets = [trainer.static_buffers(B, H, W, slot=s) for s in (0, 1)]
        self.cs = copy_stream if copy_stream is not None else pick_copy_stream(trainer.device)
        self.copied = [torch.cuda.Event() for _ in range(2)]
        self.consumed = [torch.cuda.Event() for _ in range(2)]
        cur = torch.cuda.current_stream()
        for e in self.consumed:
            e.record(cur)
        self.n_in = 0            # batches whose copy has been enqueued
        self.n_run = 0           # batches whose step has been enqueued

    def _copy(self, host_batch):
        s = self.n_in & 1
        with torch.cuda.stream(self.cs):
            self.cs.wait_event(self.consumed[s])
            for k, v in host_batch.items():
                self.sets[s][k].copy_(v, non_blocking=True)
            self.copied[s].record(self.cs)
        self.n_in += 1

    def _run(self, **kw):
        s = self.n_run & 1
        cur = torch.cuda.current_stream()
        cur.wait_event(self.copied[s])
        loss = self.tr.step(self.sets[s], **kw)
        self.consumed[s].record(cur)
        self.n_run += 1
        return loss

    def step(self, host_batch, **kw):
        self._copy(host_batch)
        if self.n_in - self.n_run < 2:
            return None
        return self._run(**kw)

    def flush(self, **kw):
        return self._run(**kw) if self.n_run < self.n_in else None
