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


class RegionFeed:
    """Variable-size census-region batches from a ``DataLoader`` to the device, AHEAD of the training step (round 6; the reference's loop
    copies every batch synchronously and then runs ~35 small launches of augmentation / normalisation on it: run_train.py:185-187,
    utils/utils.py:22-43,130-214).

        for sample in RegionFeed(loader, device):      # dicts like the loader's, tensors already on the device
            trainer.train_step(sample)

    A producer THREAD pulls the batches (with ``num_workers = 0`` the dataset access and the collate run in it: torch's copies release the
    GIL; worker processes cost more in IPC than an in-memory collate takes -- tools/feed_probe.py: 2.7 - 5.2 ms per batch through 8 workers
    with / without the loader's pinning thread against 0.7 ms for the collate itself), stages their tensors in a ring of pinned host
    buffers (grow-only) and enqueues the copies on a copy stream that does not share a hardware queue with the compute stream
    (``pick_copy_stream``); the consumer waits on the copy's event and steps.  Region shapes never recur, so the device tensors come from
    torch's caching allocator (``record_stream`` keeps a block alive until the step that reads it has run).  Non-tensor items pass
    through.  Determinism: the FIRST batch is pulled on the caller's thread -- that is where a shuffling sampler draws its seed from the
    global generator -- so the producer never touches the generators the augmentation coins and selection grids come from."""

    TENSOR_KEYS = ("S2", "S1", "admin_mask", "y", "census_idx", "building_counts")
    DEPTH = 3                       # staged batches in flight (ring slots of pinned memory)

    def __init__(self, loader, device, copy_stream=None):
        self.loader = loader
        self.device = torch.device(device)
        self.cs = copy_stream if copy_stream is not None else pick_copy_stream(self.device)
        self._pin = [{} for _ in range(self.DEPTH)]      # per slot: key -> pinned staging tensor (grow-only)
        self._done = [None] * self.DEPTH                 # per slot: event behind the slot's last copy
        self._n = 0

    def __len__(self):
        return len(self.loader)

    def _slot(self):
        slot = self._n % self.DEPTH
        self._n += 1
        if self._done[slot] is not None:
            self._done[slot].synchronize()             # DEPTH batches back: guards the staging buffers of this slot
        return slot

    def _alloc(self, slot):
        """(key, shape, dtype) -> a view of this slot's pinned staging buffer of that key (grow-only)."""
        def alloc(key, shape, dtype):
            n = 1
            for d in shape:
                n *= int(d)
            buf = self._pin[slot].get(key)
            if buf is None or buf.numel() < n or buf.dtype != dtype:
                buf = self._pin[slot][key] = torch.empty(max(int(n * 1.25), 1), dtype=dtype).pin_memory()
            return buf[:n].view(shape)
        return alloc

    def _stage(self, batch, consumer_stream, slot=None):
        if slot is None:
            slot = self._slot()
        dev, host = {}, {}
        for k, v in batch.items():
            if not (torch.is_tensor(v) and k in self.TENSOR_KEYS):
                host[k] = v
                continue
            if k in ("admin_mask", "y") and v.dtype != torch.float32:
                v = v.float()
            if not v.is_pinned():
                p = self._alloc(slot)(k, tuple(v.shape), v.dtype)
                p.copy_(v)
                v = p
            host[k] = v
        with torch.cuda.stream(self.cs):
            for k in self.TENSOR_KEYS:
                if k in host and torch.is_tensor(host[k]):
                    t = host.pop(k).to(self.device, non_blocking=True)
                    t.record_stream(consumer_stream)
                    dev[k] = t
            ev = torch.cuda.Event()
            ev.record(self.cs)
        self._done[slot] = ev
        return dev, host, ev

    def __iter__(self):
        import queue
        import threading
        from .collate import Population_Dataset_collate_fn, collate_into
        it = iter(self.loader)
        # Fast path (in-process loader with the standard collate): the index batches come from the loader's OWN iterator (same generator
        # consumption, same order), the items are assembled straight into the pinned staging slot (collate_into) -- one pass over the
        # data instead of collate + staging copy.  Anything else (worker processes, another collate): the loader's batches as they come.
        direct = (getattr(self.loader, "num_workers", 1) == 0 and getattr(self.loader, "collate_fn", None) is Population_Dataset_collate_fn and
                  hasattr(it, "_next_index") and getattr(self.loader, "batch_sampler", None) is not None)
        ds = self.loader.dataset

        def pull():
            if not direct:
                return next(it, None)
            try:
                return it._next_index()
            except StopIteration:
                return None
        first = pull()                                  # on the caller's thread: the sampler's seed draw (see the class docstring)
        if first is None:
            return
        cur = torch.cuda.current_stream(self.device)
        q = queue.Queue(maxsize=self.DEPTH - 1)
        stop = threading.Event()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()

        def produce():
            try:
                torch.cuda.set_device(dev_index)
                batch = first
                while batch is not None and not stop.is_set():
                    if direct:
                        slot = self._slot()
                        q.put(self._stage(collate_into([ds[i] for i in batch], self._alloc(slot)), cur, slot))
                    else:
                        q.put(self._stage(batch, cur))
                    batch = pull()
                q.put(None)
            except BaseException as ex:                 # surfaces in the consumer
                q.put(ex)
        th = threading.Thread(target=produce, name="popcorn-region-feed", daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                dev, host, ev = item
                cur.wait_event(ev)
                yield {**host, **dev}
        finally:
            stop.set()
            while th.is_alive():                        # a consumer that stops early: drain so that the producer can finish
                try:
                    q.get(timeout=0.05)
                except queue.Empty:
                    pass
            th.join()
