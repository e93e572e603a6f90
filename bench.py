#!/usr/bin/env python3
"""bench.py -- training throughput of the POPCORN hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: the driver launches this file once per GPU through torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the
env, backend nccl = RCCL).  Started by hand with --gpus N > 1 and no WORLD_SIZE, it starts that very launcher as a CHILD
process before anything touches the GPU and relays the child's JSON line and exit code.  A world size that differs from
--gpus is an error (exit 2), never a silent single-GPU run.

One "step" = one full optimisation step of the reference recipe (run_train.py:186-238) on a batch of synthetic
15-band 100x100 tiles already resident in HBM: band select + normalise -> frozen building-extractor U-Net ->
sparsity mask -> trainable dual-stream U-Net forward -> sparse head -> log-L1 loss + scale regulariser -> backward
(head, U-Net dgrad/wgrad) -> [N > 1: one RCCL all-reduce of the flat gradient] -> clip_grad_norm_(0.01) -> Adam.

Timing: W warm-up steps, then `--repeats` blocks of EXACTLY K steps, each block bracketed by barrier +
torch.cuda.synchronize() on both sides and reduced with MAX over ranks; `value` comes from the MEDIAN block (all block
times are in `ms_per_step_blocks`).  Rank 0 prints ONE JSON line with
  roofline            the dominant single kernel (head backward), algorithmic flops (37,376 flop per
                      selected pixel) / live event-timed launch duration, plus the executed-MFMA view (the kernel
                      recomputes the forward: 56,064 flop per pixel are issued)
  roofline_conv       the most frequent heavy launch (grouped conv 8->8 @128x128), HBM view + MFMA view
  roofline_conv_class every conv / transposed-conv launch of one step timed live (events between the eager launches):
                      per-layer us, algorithmic flops and compulsory bytes, and the class totals against both peaks
  cpu_baseline        the CPU oracle ("port") timed on the host cores, several bounded legs (N = 1 only)
  per_rank_ms_per_step  every rank's own median block (a straggler shows here; `ms_per_step` is the max over ranks)
  bf16                BASELINE config 4's mixed precision on the same workload in the same process (value, ms_per_step, HBM view
                      of its conv class, final loss) -- a second figure, never the headline
  soak                N = 1: ONE block of 1500 replayed steps (thermal / power steady state next to the 30-step blocks)
  h2d                 the same step with every batch coming from PINNED HOST memory through a copy stream, double-buffered
                      (6 pre-selected bands and the full 15-band tile): tiles/s with feeding (the headline keeps its inputs resident)
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_TRAIN_PER_TILE = 1_732_423_680          # SURVEY.md section 8d / BASELINE.md section 2 (all 10^4 px selected)
FLOP_HEAD_FWD_PX = 18_688                    # 2 x 9344 MACs per selected pixel
FLOP_HEAD_BWD_PX = 37_376                    # "head bwd 4 x 93.44 M" per 10^4-px tile = data + weight gradients
FP32_MATRIX_PEAK = 157.3e12                  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
BF16_MATRIX_PEAK = 2500.0e12                 # MI355X_MICROARCH.md: v_mfma_f32_16x16x32_bf16, dense (bf16 mode only)
HBM_PEAK = 8.0e12                            # MI355X_MICROARCH.md: HBM3E spec (6.3 TB/s is what a float4 copy achieves)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=7, help="timed blocks of --steps steps; the median block is reported")
    ap.add_argument("--batch", type=int, default=64, help="tiles per GPU per step (BASELINE config: 64)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP-graph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-class-sweep", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=7.0, help="CPU work per cpu_baseline leg")
    ap.add_argument("--precision", choices=["fp32", "bf16"], default="fp32",
                    help="bf16: BASELINE config 4's mixed precision (PC_PREC_BF16) as a SECOND line; the headline stays fp32")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra objects of the line: `bf16` (config 4's precision, same process), `soak` (N = 1: one "
                         "1500-step block) and `h2d` (the step fed from pinned host memory through a copy stream)")
    ap.add_argument("--extras-multi", action="store_true",
                    help="N > 1: also run the `bf16` and `h2d` legs (default: skipped, the multi-GPU run stays under two minutes)")
    ap.add_argument("--no-config-legs", action="store_true", help="skip the N = 1 legs `fwd_parity`, `grad_parity`, `batch16`, `config3_regions`, `config3_epoch`, `config5`")
    ap.add_argument("--soak-steps", type=int, default=1500)
    ap.add_argument("--prewarm-seconds", type=float, default=3.0,
                    help="untimed steps in front of the W warm-up steps until this much wall time has passed: the part's clocks settle "
                         "into their sustained state only after ~3 s of load (30-step blocks right after start-up read 1.81 ms, the same "
                         "blocks after 3 s 1.78 ms, a 1500-step block 1.77 ms); 0 = off")
    return ap.parse_args()


# ---- N > 1 without a launcher: become the launcher's parent (no GPU call has happened yet) ---------------------------
def relaunch_distributed(args):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


# ---- roofline objects -------------------------------------------------------------------------------------------------
def _pmc(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return None


class PowerSampler:
    """Board power and shader clock from the hwmon files of the CURRENT device (matched by PCI address), sampled by a thread every
    20 ms between start() and stop(): what the part draws and clocks at while a leg runs.  Silent (all None) where sysfs is not
    readable."""

    def __init__(self, torch):
        import glob
        self.fpower = self.fsclk = None
        try:
            pr = torch.cuda.get_device_properties(torch.cuda.current_device())
            bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            for card in glob.glob("/sys/class/drm/card*"):
                if os.path.realpath(os.path.join(card, "device")).endswith(bdf):
                    hw = glob.glob(os.path.join(card, "device", "hwmon", "hwmon*"))
                    if hw:
                        for name in ("power1_input", "power1_average"):
                            if os.path.exists(os.path.join(hw[0], name)):
                                self.fpower = os.path.join(hw[0], name)
                                break
                        if os.path.exists(os.path.join(hw[0], "freq1_input")):
                            self.fsclk = os.path.join(hw[0], "freq1_input")
        except Exception:
            pass
        self.samples, self._stop, self._th = [], False, None

    @staticmethod
    def _read(path):
        try:
            with open(path) as fh:
                return float(fh.read().strip())
        except Exception:
            return None

    def start(self):
        import threading
        self.samples, self._stop = [], False

        def loop():
            while not self._stop:
                self.samples.append((self._read(self.fpower) if self.fpower else None, self._read(self.fsclk) if self.fsclk else None))
                time.sleep(0.02)
        self._th = threading.Thread(target=loop, daemon=True)
        self._th.start()

    def stop(self):
        self._stop = True
        if self._th:
            self._th.join()
        pw = [p for p, _ in self.samples[2:] if p]
        ck = [c for _, c in self.samples[2:] if c]
        return {"board_power_w": round(statistics.mean(pw) / 1e6, 0) if pw else None,
                "sclk_mhz": round(statistics.mean(ck) / 1e6, 0) if ck else None,
                "sclk_mhz_min": round(min(ck) / 1e6, 0) if ck else None, "samples": len(self.samples)}


def _spin_cycles(torch, us=150.0):
    """torch.cuda._sleep argument for ~`us` microseconds of device-side spinning (calibrated once)."""
    if not hasattr(_spin_cycles, "per_us"):
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(1000)
        torch.cuda.synchronize()
        c0.record(); torch.cuda._sleep(2_000_000); c1.record()
        torch.cuda.synchronize()
        _spin_cycles.per_us = 2_000_000 / max(c0.elapsed_time(c1) * 1e3, 1e-3)
    return int(_spin_cycles.per_us * us)


class _LocalReducer:
    """Stand-in for FlatReducer inside the roofline legs: they run on rank 0 only, so no collective may be part of their steps (the
    timed kernel does not depend on them)."""
    active, capture_failed, world = False, False, 1

    def reduce_stats(self, t):
        return t

    def reduce_grads(self, t):
        return t

    def global_batch(self, b):
        return b

    def capturable(self):
        return False


def _head_bwd_in_replayed_step(torch, trainer, sample, reps):
    """Median seconds of the head-backward call (pack + kernel + reduce launches) inside replayed train steps: see the caller."""
    from popcorn_amd import ops, _lib as L
    dkey = "input" if sample.get("input") is not None else "raw"
    B, _, H, W = sample[dkey].shape
    st = {k: v for k, v in sample.items() if not k.startswith("_")}
    sel = trainer._draw_selection(H, W).to(sample[dkey].device)
    snap = (trainer.flat_p.clone(), trainer.m.clone(), trainer.v.clone(), trainer.step_count.clone())
    saved_reducer = trainer.reducer
    graphs = [torch.cuda.CUDAGraph() for _ in range(3)]
    orig = ops.head_bwd
    pool = torch.cuda.graph_pool_handle()
    active = [None]                    # index of the graph whose capture is open (so that a failure can always close it)

    def begin(i):
        graphs[i].capture_begin(pool=pool)
        active[0] = i

    def end(i):
        graphs[i].capture_end()
        active[0] = None

    def split(*a, **k):
        end(0)
        begin(1)
        r = orig(*a, **k)
        end(1)
        begin(2)
        return r

    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    ops.head_bwd = split
    trainer.reducer = _LocalReducer()
    try:
        with torch.cuda.stream(side), L.precision(trainer.model.precision):
            try:
                begin(0)
                trainer._forward(st, sel, False, False)
                trainer._backward(st, False, False)
                trainer._update(False, False)
                end(2)
            finally:
                if active[0] is not None:            # never leave the stream in capture mode
                    try:
                        graphs[active[0]].capture_end()
                    except Exception:
                        pass
                    active[0] = None
    finally:
        ops.head_bwd = orig
        trainer.reducer = saved_reducer
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps + 3)]
    for e0, e1 in ev:
        graphs[0].replay()
        e0.record()
        graphs[1].replay()
        e1.record()
        graphs[2].replay()
    torch.cuda.synchronize()
    trainer.flat_p.copy_(snap[0]); trainer.m.copy_(snap[1]); trainer.v.copy_(snap[2]); trainer.step_count.copy_(snap[3])
    del graphs
    return statistics.median([e0.elapsed_time(e1) for e0, e1 in ev[3:]]) * 1e-3


def head_bwd_roofline(torch, trainer, sample, reps=10):
    """`roofline_head` (round 6: `roofline` is the fused conv backward, the top row of the tracked kernel stats by share).
    `head_bwd_pc_kernel`, the largest SINGLE launch of the step (profiles/r*_kernel_stats.csv): backward of the
    sparse 16-64-64-64-1 head.  Timed live between two events on the launch stream, inside `reps` eager train steps on the
    step's real tensors (the call = weight-image pack + the kernel + its small reduce launch).  ALGORITHMIC flops per launch = 37,376 per
    selected pixel (data + weight gradients).  The kernel also recomputes the forward chain in registers
    instead of reading a 491 MB hidden-activation buffer: it ISSUES 56,064 flop per pixel -- reported separately as the
    executed-MFMA fraction, never as `frac`."""
    from popcorn_amd import ops, _lib as L
    m = trainer.model
    X = sample["input"]
    B, _, H, W = X.shape
    bf = m.precision == "bf16"
    peak = BF16_MATRIX_PEAK if bf else FP32_MATRIX_PEAK
    esz = 2 if bf else 4
    nsel = B * H * W                      # bench regions cover the tile: every pixel is selected

    # Timed INSIDE the train step (eager launches of the whole step, the head-backward call between two events behind a device-side
    # spin so that the host's launch preparation is not in the figure): the call then sees the step's real tensors, mask and cache
    # state.  (Back-to-back calls on the same tensors -- the earlier form of this leg -- read 285 us where the step's kernel takes
    # 321 us under rocprofv3: the 41 MB feature map stays in the Infinity Cache between repetitions.)
    spin = _spin_cycles(torch)
    rec = []
    orig = ops.head_bwd

    def timed(*a, **k):
        torch.cuda._sleep(spin)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        r = orig(*a, **k)
        t1.record()
        rec.append((t0, t1))
        return r
    saved_graph, saved_reducer = trainer.use_graph, trainer.reducer
    snap = (trainer.flat_p.clone(), trainer.m.clone(), trainer.v.clone(), trainer.step_count.clone())
    ops.head_bwd = timed
    trainer.use_graph = False
    trainer.reducer = _LocalReducer()
    trainer._native_ok = lambda s: False      # the per-launch engine (same kernels): the native executor never passes through ops.head_bwd
    try:
        for r in range(reps + 2):
            torch.manual_seed(1)
            trainer.step(sample)
        torch.cuda.synchronize()
    finally:
        ops.head_bwd = orig
        del trainer._native_ok
        trainer.use_graph, trainer.reducer = saved_graph, saved_reducer
        trainer.flat_p.copy_(snap[0]); trainer.m.copy_(snap[1]); trainer.v.copy_(snap[2]); trainer.step_count.copy_(snap[3])
    dur_eager = statistics.median([t0.elapsed_time(t1) for t0, t1 in rec[2:]]) * 1e-3
    # ... and under GRAPH REPLAY, which is what `value` runs: the step captured as three graphs split at the head-backward call
    # (before | the call | after, one memory pool), replayed back to back with an event on either side of the middle one.  The
    # device never idles (the host is a step ahead), so this is the kernel in the state the timed region has it in; it reads ~10 %
    # above the eager figure and agrees with rocprofv3's per-kernel average of the replayed step (profiles/).
    dur_graph = None
    try:
        dur_graph = _head_bwd_in_replayed_step(torch, trainer, sample, reps)
    except Exception as ex:                                  # pragma: no cover  (measurement nicety: the eager figure stands in)
        print(f"bench.py: split-graph timing of the head backward failed ({type(ex).__name__}: {ex}); eager figure used", file=sys.stderr)
        torch.cuda.synchronize()
    dur = dur_graph if dur_graph else dur_eager
    flops = float(FLOP_HEAD_BWD_PX) * nsel
    issued = float(FLOP_HEAD_BWD_PX + FLOP_HEAD_FWD_PX) * nsel
    achieved = flops / dur / 1e12
    pmc = _pmc("r6_pmc_head_bwd.json") or _pmc("r5_pmc_head_bwd.json") or _pmc("r4_pmc_head_bwd.json") or _pmc("r3_pmc_head_bwd.json") or _pmc("r2_pmc_head_bwd.json")
    traffic = pmc["traffic_bytes"] if pmc and (B, H, W) == (64, 100, 100) else None
    split = (not bf) and bool(L.lib().pc_get_head_split())
    rp_us, rp_file, rp_match = _rocprof_avg_us("head_bwd_bf16_coop4_kernel" if bf else ("head_bwd_pc_kernel<0, true>" if split else "head_bwd_pc_kernel<0, false>"),
                                               "bf16" if bf else "fp32")
    kname = ("head_bwd_bf16_coop4_kernel (sparse head backward, bf16 MFMA 16x16x32 / 16x16x16, the 4 waves of a workgroup share the weight gradients through an LDS exchange + transposing reads, 2 workgroups per CU)" if bf else
             ("head_bwd_pc_kernel<0, true> (sparse head backward, producer/consumer waves; fp32 results through exact 3-way bf16 operand "
              "splits: six bf16 partial products per fp32 product on v_mfma_f32_16x16x32_bf16 / 16x16x16_bf16, fp32 accumulation -- measured "
              "against float64 at the accuracy of the fp32-MFMA form, tests/test_gpu_convt_head.py)" if split else
              "head_bwd_pc_kernel<0, false> (sparse head backward, producer/consumer waves, fp32 MFMA 16x16x4)"))
    if bf:
        pmc = (_pmc("r6_pmc_head_bwd_bf16.json") or _pmc("r5_pmc_head_bwd_bf16.json") or _pmc("r4_pmc_head_bwd_bf16.json") or _pmc("r3_pmc_head_bwd_bf16.json") or
               _pmc("r2_pmc_head_bwd_bf16.json"))
        traffic = pmc["traffic_bytes"] if pmc and (B, H, W) == (64, 100, 100) else None
    base = {"bound": "mfma", "kernel": kname + " + the reduce launch of the same call", "unit": "TFLOP/s", "traffic": traffic,
            "launch_us": round(dur * 1e6, 2), "launch_us_eager_step": round(dur_eager * 1e6, 2),
            # STATIC fields: the tracked rocprofv3 average of this kernel, reported only when the tracked profile was collected from THIS
            # tree's kernel sources (stamp); the live measurement of the run is `launch_us` / `frac`
            "rocprof_us": rp_us, "rocprof_file": rp_file, "rocprof_stamp_matches_tree": rp_match,
            "timed": ("graph replay of the step split at the call (events around the middle graph)" if dur_graph else
                      "eager steps, events around the call behind a device-side spin"),
            "alg_flop_per_launch": flops, "units_per_launch": nsel,
            "unit_def": f"selected pixel, {FLOP_HEAD_BWD_PX} flop (SURVEY.md 8d head backward)",
            "alg_bytes_per_launch": nsel * (16 * esz + 4 + 4 + 1) + B * 16 * (H + 28) * (W + 28) * esz}
    if split:
        # The split form runs on the bf16 matrix pipe: `achieved` / `peak` / `frac` are what THAT pipe executes (six bf16 partial products
        # per fp32 product, the forward chain recomputed in registers included) against ITS dense peak -- never a figure above 1 against a
        # peak the kernel does not run on.  The algorithmic fp32 work is reported beside it, against the method's
        # own ceiling (bf16 peak / 6) and, for continuity with rounds 1-5, against the fp32 matrix peak.
        executed = 6 * issued
        base.update({"pipe": "bf16", "achieved": round(executed / dur / 1e12, 3), "peak": BF16_MATRIX_PEAK / 1e12,
                     "frac": round(executed / dur / BF16_MATRIX_PEAK, 4), "pipe_frac": round(executed / dur / BF16_MATRIX_PEAK, 4),
                     "executed_flop_per_launch": executed,
                     "alg_tflops": round(achieved, 3), "alg_ceiling_tflops": round(BF16_MATRIX_PEAK / 6 / 1e12, 1),
                     "alg_frac_of_ceiling": round(achieved * 1e12 / (BF16_MATRIX_PEAK / 6), 4),
                     "alg_frac_of_fp32_matrix_peak": round(achieved * 1e12 / FP32_MATRIX_PEAK, 4),
                     "rocprof_frac": (round(executed / (rp_us * 1e-6) / BF16_MATRIX_PEAK, 4) if rp_us and (B, H, W) == (64, 100, 100) else None),
                     "rocprof_alg_frac_of_fp32_matrix_peak": (round(flops / (rp_us * 1e-6) / FP32_MATRIX_PEAK, 4)
                                                               if rp_us and (B, H, W) == (64, 100, 100) else None)})
    else:
        base.update({"pipe": "bf16" if bf else "fp32", "achieved": round(achieved, 3), "peak": peak / 1e12,
                     "frac": round(achieved * 1e12 / peak, 4),
                     "rocprof_frac": (round(flops / (rp_us * 1e-6) / peak, 4) if rp_us and (B, H, W) == (64, 100, 100) else None),
                     "executed_mfma_view": {"flop_per_unit": FLOP_HEAD_BWD_PX + FLOP_HEAD_FWD_PX, "achieved_tflops": round(issued / dur / 1e12, 3),
                                            "frac": round(issued / dur / peak, 4),
                                            "note": "includes the forward chain recomputed in registers (not algorithmic work)"}})
    return base


def fused_conv_bwd_roofline(torch, trainer, sample, reps=8):
    """`roofline` (fp32 line, round 6): the kernel with the LARGEST SHARE of the step in the tracked rocprofv3 summary --
    `conv3x3_bwd_s3_kernel<8, false>`, data + weight + bias gradient of an 8 -> 8 conv layer in one launch (five launches per step: up1b,
    the skip block of up1a, inc2 at 128 x 128, up2b and the two skip halves of up2a at 64 x 64; networks.py:259-266,318 backward).  With its
    operands split into three bf16 planes (6 x 16 matrix cycles per 32 K-slots) the kernel sits BELOW the ridge of the pipe it runs on
    (24 flop per byte against 2517 / 6 / 8 = 52): bound = hbm.  `achieved` = ALGORITHMIC bytes of the step's five launches (every distinct
    gradient / input / output tensor once) / their summed durations, timed live: eager train steps through the per-launch engine (same
    kernels as the captured step), two events around every launch behind a device-side spin.  `mfma_view`: the same launches as matrix
    work (executed = 6 partial products per product + the bias-gradient column, against the bf16 dense peak; algorithmic against the
    method's ceiling bf16 / 6)."""
    from popcorn_amd import ops, _lib as L
    if trainer.model.precision != "fp32" or not L.lib().pc_get_conv_split():
        return None
    spin = _spin_cycles(torch)
    rec = []
    orig = ops.WgradBatch.conv3x3_bwd_group

    def timed(self, problems, cin_total, c0, accumulate=False):
        g0 = problems[0]["g"]
        mine = g0.shape[1] == 8 and problems[0].get("pool_act") is None and g0.dtype == torch.float32
        if not mine:
            return orig(self, problems, cin_total, c0, accumulate=accumulate)
        uniq = {}
        for pr in problems:
            for k in ("g", "x", "out"):
                uniq[pr[k].data_ptr()] = pr[k].numel() * 4
        B, _, H, W = g0.shape
        torch.cuda._sleep(spin)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        r = orig(self, problems, cin_total, c0, accumulate=accumulate)
        t1.record()
        rec.append((t0, t1, sum(uniq.values()), len(problems) * B * H * W, f"{len(problems)}x{B}x{H}x{W}"))
        return r
    saved_graph, saved_reducer = trainer.use_graph, trainer.reducer
    snap = (trainer.flat_p.clone(), trainer.m.clone(), trainer.v.clone(), trainer.step_count.clone())
    ops.WgradBatch.conv3x3_bwd_group = timed
    trainer.use_graph = False
    trainer.reducer = _LocalReducer()
    trainer._native_ok = lambda s: False      # the per-launch engine (same kernels, same order as the native executor)
    try:
        for r in range(reps + 2):
            torch.manual_seed(1)
            trainer.step(sample)
        torch.cuda.synchronize()
    finally:
        ops.WgradBatch.conv3x3_bwd_group = orig
        del trainer._native_ok
        trainer.use_graph, trainer.reducer = saved_graph, saved_reducer
        trainer.flat_p.copy_(snap[0]); trainer.m.copy_(snap[1]); trainer.v.copy_(snap[2]); trainer.step_count.copy_(snap[3])
    per_step = len(rec) // (reps + 2)
    if per_step == 0:
        return None
    rec = rec[2 * per_step:]
    steps = [rec[i * per_step:(i + 1) * per_step] for i in range(reps)]
    tot = [sum(t0.elapsed_time(t1) for t0, t1, *_ in st) * 1e-3 for st in steps]
    dur = statistics.median(tot)                                      # seconds for the step's launches of this kernel
    nbytes = float(sum(b for _, _, b, _, _ in steps[0]))
    px = float(sum(n for *_, n, _ in steps[0]))
    alg_flop = px * 2.0 * (2 * 9 * 8 * 8)                             # data + weight gradient, 2 flop per MAC
    # executed on the bf16 pipe per 128-pixel strip: 150 instructions of 16 x 16 x 32 (72 data gradient incl. the zero tap plane, 72 weight
    # gradient, 6 bias column) = 150 * 16384 flop
    executed = px / 128.0 * 150.0 * 16384.0
    pmc = _pmc("r6_pmc_conv_bwd.json")
    B, _, H, W = sample["input"].shape
    traffic = pmc["traffic_bytes"] if pmc and (B, H, W) == (64, 100, 100) else None
    rp_us, rp_file, rp_match = _rocprof_avg_us("conv3x3_bwd_s3_kernel<8, false>", "fp32")
    n = per_step
    return {"bound": "hbm",
            "kernel": "conv3x3_bwd_s3_kernel<8, false> (data + weight + bias gradient of an 8 -> 8 conv layer in one launch; fp32 tensors, operands "
                      "split exactly into three bf16 planes once per strip while it is staged into LDS, six partial products per product on "
                      "v_mfma_f32_16x16x32_bf16, fp32 accumulation) -- the top row of the tracked kernel stats by share: "
                      f"{n} launches per step ({', '.join(lbl for *_, lbl in steps[0])})",
            "achieved": round(nbytes / dur / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(nbytes / dur / HBM_PEAK, 4),
            "traffic": traffic, "launches_per_step": n,
            "launch_us": round(dur / n * 1e6, 2), "alg_bytes_per_launch": nbytes / n, "alg_flop_per_launch": alg_flop / n,
            "units_per_launch": px / n, "unit_def": "pixel of an 8 <-> 8 layer: 96 algorithmic bytes (gradient + input + data gradient, 8 fp32 "
                                                    "channels each), 2,304 flop (SURVEY.md 8d: 1,152 per pixel and pass)",
            "rocprof_us": rp_us, "rocprof_file": rp_file, "rocprof_stamp_matches_tree": rp_match,
            "rocprof_frac": (round(nbytes / n / (rp_us * 1e-6) / HBM_PEAK, 4) if rp_us and (B, H, W) == (64, 100, 100) else None),
            "timed": "eager steps of the per-launch engine, events around every launch of the kernel behind a device-side spin; mean launch of the step",
            "mfma_view": {"pipe": "bf16", "executed_tflops": round(executed / dur / 1e12, 2), "peak": BF16_MATRIX_PEAK / 1e12,
                          "pipe_frac": round(executed / dur / BF16_MATRIX_PEAK, 4), "alg_tflops": round(alg_flop / dur / 1e12, 2),
                          "alg_ceiling_tflops": round(BF16_MATRIX_PEAK / 6 / 1e12, 1),
                          "alg_frac_of_ceiling": round(alg_flop / dur / (BF16_MATRIX_PEAK / 6), 4),
                          "flop_per_byte": round(alg_flop / nbytes, 1), "ridge_flop_per_byte": round(BF16_MATRIX_PEAK / 6 / HBM_PEAK, 1)}}


def conv_kernel_roofline(torch, B, reps=5, nsets=4):
    """The most frequent heavy launch of the step, the grouped 3x3 conv 8->8 @128x128 (4 problems = 2 networks x 2
    streams, B tiles each), timed live with events over rotating buffer sets (cold Infinity Cache).  It sits at the ridge
    of this chip (18 flop / compulsory byte): algorithmic bytes against the HBM peak and algorithmic flops against the
    fp32-matrix peak."""
    from popcorn_amd import ops, _lib as L
    from popcorn_amd.train import _capturing
    adt = L.act_dtype()
    esz = 2 if adt == torch.bfloat16 else 4
    peak = BF16_MATRIX_PEAK if esz == 2 else FP32_MATRIX_PEAK
    sets = []
    for _ in range(nsets):
        probs = []
        for _ in range(4):
            bias = torch.zeros(8, device="cuda")
            probs.append({"a": L.as_act(torch.randn(B, 8, 128, 128, device="cuda")), "w": torch.randn(8, 8, 3, 3, device="cuda") * 0.1,
                          "bn": L.bn(bias), "out": L.empty_act(B, 8, 128, 128, "cuda"), "_keep": bias})
        sets.append(probs)
    for s in sets:
        ops.conv3x3_fwd_group(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream()
    with torch.cuda.stream(cap):
        with _capturing(g, stream=cap):         # (cyclic GC off during the capture: popcorn_amd/train.py)
            for _ in range(reps):
                for s in sets:
                    ops.conv3x3_fwd_group(s)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    dur = e0.elapsed_time(e1) * 1e-3 / (reps * nsets)
    nbytes = 4 * B * 128 * 128 * esz * (8 + 8)               # compulsory: read 8 channels, write 8 channels
    flops = 4 * B * 128 * 128 * 2 * 9 * 8 * 8
    pmc = (_pmc("r6_pmc_conv_8to8_bf16.json" if esz == 2 else "r6_pmc_conv_8to8.json") or
           _pmc("r5_pmc_conv_8to8_bf16.json" if esz == 2 else "r5_pmc_conv_8to8.json") or
           _pmc("r4_pmc_conv_8to8_bf16.json" if esz == 2 else "r4_pmc_conv_8to8.json") or
           _pmc("r3_pmc_conv_8to8_bf16.json" if esz == 2 else "r3_pmc_conv_8to8.json"))
    split = esz == 4 and int(L.lib().pc_get_conv_split()) != 0
    kname = ("conv3x3_cl_kernel<8,8,fwd> (channels-last bf16)" if esz == 2 else
             ("conv3x3_fwd_s3_kernel<8,8> (fp32 tensors, split operands on the bf16 matrix pipe)" if split else "conv3x3_mfma_kernel<8,8,fwd> (fp32)"))
    return {"bound": "hbm", "kernel": f"{kname} grouped x4 (3x3 conv + BN + ReLU, 8->8 @128x128)",
            "achieved": round(nbytes / dur / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(nbytes / dur / HBM_PEAK, 4),
            "traffic": pmc["traffic_bytes"] if pmc and B == 64 else None, "launch_us": round(dur * 1e6, 2),
            "alg_bytes_per_launch": nbytes,
            "mfma_view": {"achieved_tflops": round(flops / dur / 1e12, 2), "peak": peak / 1e12,
                          "frac": round(flops / dur / peak, 4)}}


def conv_class_sweep(torch, trainer, sample, reps=3):
    """Every conv / transposed-conv launch of ONE train step, timed live: the step runs eagerly with an event recorded
    after each launch of the class (elapsed between consecutive events = that launch on the busy device).  Per launch:
    us, algorithmic flops (2 * taps * Cin * Cout per output pixel per problem) and compulsory bytes (every operand tensor
    read once, every result written once, fp32).  Class totals against the fp32-matrix peak and the HBM peak."""
    from popcorn_amd import ops
    rec, names = [], []
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731

    spin_cycles = _spin_cycles(torch)

    def numel(t):
        return 0 if t is None else t.numel()

    def nb(t):
        return 0 if t is None else t.numel() * t.element_size()

    def wrap(obj, attr, label, cost):
        orig = getattr(obj, attr)

        def fn(*a, **k):
            # a device-side delay in front of the first event: the host builds the launch (descriptors, ctypes call: 20-40 us)
            # while the device spins, so that start event, kernel(s) and end event are all queued when the device reaches them
            # -- without it the device idles between the two events while the host prepares the launch, and a 18 us kernel
            # reads 42 us
            torch.cuda._sleep(spin_cycles)
            e0 = ev(); e0.record()
            r = orig(*a, **k)
            e1 = ev(); e1.record()
            fl, by, desc = cost(*a, **k)
            rec.append((label, desc, e0, e1, fl, by))
            return r
        setattr(obj, attr, fn)
        return (obj, attr, orig)

    def c_fwd(problems, **k):
        p0 = problems[0]
        cout, cin = p0["w"].shape[0], p0["w"].shape[1]
        o = p0.get("out") if p0.get("out") is not None else p0["dot_out"]
        Bn, _, H, W = o.shape
        n = len(problems)
        px = Bn * H * W
        by = 0
        for p in problems:
            by += nb(p["a"]) * (cin if k.get("a_channels") else p["a"].shape[1]) // p["a"].shape[1] + nb(p.get("b"))
            by += nb(p.get("out")) + nb(p.get("pool_out")) + (nb(p.get("dot_out")) if p.get("dot_w") is not None else 0)
        return 2.0 * 9 * cin * cout * px * n, by, f"fwd {cin}->{cout} @{H}x{W} x{n}"

    def c_dgrad(problems, c0, cn, pool=False, accumulate=False):
        p0 = problems[0]
        Bn, cg, H, W = p0["g"].shape
        n = len(problems)
        by = sum(nb(p["g"]) + nb(p["out"]) * (2 if (accumulate or pool) else 1) + nb(p.get("act")) for p in problems)
        return 2.0 * 9 * cg * cn * Bn * H * W * n, by, f"dgrad {cg}->{cn} @{H}x{W} x{n}{' pool' if pool else ''}"

    def c_wgrad_g(self, problems, cout, **k):
        p0 = problems[0]
        Bn, cg, H, W = p0["g"].shape
        cin = (k.get("a_channels") or p0["a"].shape[1]) + (p0["b"].shape[1] if p0.get("b") is not None else 0)
        n = len(problems)
        by = sum(nb(p["g"]) + nb(p["a"]) + nb(p.get("b")) for p in problems)
        return 2.0 * 9 * cin * cout * Bn * H * W * n, by, f"wgrad {cin}->{cout} @{H}x{W} x{n}"

    def c_wgrad_1(self, a, g, cout, dw, db, **k):
        Bn, cg, H, W = g.shape
        cin = k.get("a_channels") or a.shape[1]
        return 2.0 * 9 * cin * cout * Bn * H * W, nb(g) + Bn * cin * H * W * a.element_size(), f"wgrad {cin}->{cout} @{H}x{W} x1 (reflect loader)"

    def c_bwd(self, problems, cin_total, c0, accumulate=False):
        # data + weight gradient of one layer (column block) in one launch: g and x read once, the data gradient written once
        p0 = problems[0]
        Bn, cg, H, W = p0["g"].shape
        cx = p0["x"].shape[1]
        n = len(problems)
        pool = p0.get("pool_act") is not None
        by = sum(nb(p["g"]) + nb(p["x"]) + nb(p["out"]) * (2 if (accumulate or pool) else 1) + nb(p.get("pool_act")) for p in problems)
        return 2.0 * 2 * 9 * cg * cx * Bn * H * W * n, by, f"dgrad+wgrad {cg}<->{cx} @{H}x{W} x{n}{' pool' if pool else ''}"

    def c_convt(problems):
        p0 = problems[0]
        Bn, C_, H, W = p0["x"].shape
        n = len(problems)
        return 2.0 * 4 * C_ * C_ * Bn * H * W * n, sum(nb(p["x"]) + nb(p["out"]) for p in problems), f"convT fwd {C_} @{H}x{W} x{n}"

    def c_convt_d(problems):
        p0 = problems[0]
        Bn, C_, H, W = p0["out"].shape
        n = len(problems)
        by = sum(nb(p["g"]) + nb(p["out"]) + nb(p.get("act")) for p in problems)
        return 2.0 * 4 * C_ * C_ * Bn * H * W * n, by, f"convT dgrad {C_} @{H}x{W} x{n}"

    def c_convt_w(self, problems):
        p0 = problems[0]
        Bn, C_, H, W = p0["x"].shape
        n = len(problems)
        return 2.0 * 4 * C_ * C_ * Bn * H * W * n, sum(nb(p["x"]) + nb(p["g"]) for p in problems), f"convT wgrad {C_} @{H}x{W} x{n}"

    def c_finish(self):
        return 0.0, 0, "wgrad second stage (batched reduce)"

    def c_level2(problems):
        # DoubleConv(16,16) @32x32 + ConvTranspose2d(16,16,2,2) in one launch: x read, u2 (and c1, c2 where kept) written
        p0 = problems[0]
        Bn = p0["x"].shape[0]
        n = len(problems)
        by = sum(nb(p["x"]) + nb(p["u2"]) + nb(p.get("c1")) + nb(p.get("c2")) for p in problems)
        return (2.0 * 2 * 9 * 16 * 16 + 2.0 * 4 * 16 * 16) * Bn * 32 * 32 * n, by, f"level2 fwd (2 x conv 16->16 + convT 16) @32x32 x{n}"

    def c_level2_bwd(self, problems):
        p0 = problems[0]
        Bn = p0["g2"].shape[0]
        n = len(problems)
        by = sum(nb(p["g2"]) + nb(p["c1"]) + nb(p["x"]) + nb(p["act"]) + 2 * nb(p["out"]) for p in problems)
        return 4 * 2.0 * 9 * 16 * 16 * Bn * 32 * 32 * n, by, f"level2 bwd (2 x (dgrad + wgrad) 16<->16, pool scatter) @32x32 x{n}"

    def c_up_fwd(problems, relu=True):
        # first conv of an Up block from the low-resolution map: algorithmic work = the conv over cat[skip, up-sampled] it replaces
        # (+ the transposed conv that made the up-sampled half); compulsory bytes: skip + low-res map in, output out
        p0 = problems[0]
        Bn, Cs, H, W = p0["skip"].shape
        Cz = p0["z"].shape[1]
        n = len(problems)
        by = sum(nb(p["skip"]) + nb(p["z"]) + nb(p["out"]) for p in problems)
        fl = (2.0 * 9 * (Cs + Cz) * 8 * H * W + 2.0 * 4 * Cz * Cz * (H // 2) * (W // 2)) * Bn * n
        return fl, by, f"up-conv fwd ({Cs}+{Cz}@half)->8 @{H}x{W} x{n} (conv3x3 o convT composed, no up-sampled map)"

    def c_up_bwd(self, problems):
        p0 = problems[0]
        Bn, Cg, H, W = p0["g"].shape
        Cz = p0["z"].shape[1]
        n = len(problems)
        by = sum(nb(p["g"]) + nb(p["z"]) + nb(p.get("gz")) for p in problems)
        # data + weight gradient of the up-sampled column block (2 x 2*9*Cz*8 per px) + the transposed conv's backward (2 x 2*4*Cz*Cz per low-res px)
        fl = (2 * 2.0 * 9 * Cz * 8 * H * W + 2 * 2.0 * 4 * Cz * Cz * (H // 2) * (W // 2)) * Bn * n
        return fl, by, f"up-conv bwd (up-sampled half, {Cz}@half) @{H}x{W} x{n} (one pass over the gradient)"

    def c_compose(problems):
        return 0.0, 0, f"compose conv3x3 o convT weights x{len(problems)}"

    saved_graph = trainer.use_graph
    trainer.use_graph = False
    trainer._native_ok = lambda s: False      # the per-launch engine (same kernels): the wrapped ops functions are its launches
    patches = [wrap(ops, "conv3x3_fwd_group", "conv_fwd", c_fwd), wrap(ops, "conv3x3_dgrad_group", "conv_dgrad", c_dgrad),
               wrap(ops, "level2_fwd_group", "level2_fused", c_level2),
               wrap(ops.WgradBatch, "level2_bwd_group", "level2_fused", c_level2_bwd),
               wrap(ops, "conv3x3_up_fwd_group", "up_composed", c_up_fwd), wrap(ops.WgradBatch, "up_bwd_group", "up_composed", c_up_bwd),
               wrap(ops, "conv3x3_up_compose", "up_composed", c_compose),
               wrap(ops.WgradBatch, "conv3x3_group", "conv_wgrad", c_wgrad_g), wrap(ops.WgradBatch, "conv3x3", "conv_wgrad", c_wgrad_1),
               wrap(ops.WgradBatch, "conv3x3_bwd_group", "conv_bwd_fused", c_bwd),
               wrap(ops, "convt2x2_group", "convt", c_convt), wrap(ops, "convt2x2_dgrad_group", "convt", c_convt_d),
               wrap(ops.WgradBatch, "convt2x2_group", "convt", c_convt_w), wrap(ops.WgradBatch, "finish", "conv_wgrad", c_finish)]
    # the engine module holds its own references to the ops functions through `ops.<name>` lookups: patched above
    snap = (trainer.flat_p.clone(), trainer.m.clone(), trainer.v.clone(), trainer.step_count.clone())
    try:
        per = {}
        for r in range(reps + 1):
            rec.clear()
            torch.manual_seed(1)
            trainer.step(sample)
            torch.cuda.synchronize()
            if r == 0:
                continue                                     # first eager pass: attribute queries, allocations
            for i, (label, desc, e0, e1, fl, by) in enumerate(rec):
                per.setdefault(i, [label, desc, [], fl, by])[2].append(e0.elapsed_time(e1) * 1e3)
    finally:
        for obj, attr, orig in patches:
            setattr(obj, attr, orig)
        del trainer._native_ok
        trainer.use_graph = saved_graph
        trainer.flat_p.copy_(snap[0]); trainer.m.copy_(snap[1]); trainer.v.copy_(snap[2]); trainer.step_count.copy_(snap[3])
    mpeak = BF16_MATRIX_PEAK if trainer.model.precision == "bf16" else FP32_MATRIX_PEAK
    layers, tot = [], {}
    for i in sorted(per):
        label, desc, us, fl, by = per[i]
        u = statistics.median(us)
        layers.append({"launch": desc, "us": round(u, 1), "gflop": round(fl / 1e9, 3), "mbytes": round(by / 1e6, 1),
                       "tflops": round(fl / u / 1e6, 1) if u > 0 else None, "gbps": round(by / u / 1e3, 0) if u > 0 else None})
        t = tot.setdefault(label, [0.0, 0.0, 0.0, 0])
        t[0] += u; t[1] += fl; t[2] += by; t[3] += 1
    us = sum(t[0] for t in tot.values())
    fl = sum(t[1] for t in tot.values())
    by = sum(t[2] for t in tot.values())
    return {"launches": len(layers), "us": round(us, 1), "alg_gflop": round(fl / 1e9, 2), "alg_mbytes": round(by / 1e6, 1),
            "mfma_view": {"achieved_tflops": round(fl / us / 1e6, 2), "peak": mpeak / 1e12,
                          "frac": round(fl / us * 1e6 / mpeak, 4)},
            "hbm_view": {"achieved_gbps": round(by / us / 1e3, 1), "peak": HBM_PEAK / 1e9, "frac": round(by / us * 1e6 / HBM_PEAK, 4)},
            "by_kind": {k: {"launches": t[3], "us": round(t[0], 1), "tflops": round(t[1] / t[0] / 1e6, 2), "gbps": round(t[2] / t[0] / 1e3, 1)}
                        for k, t in tot.items()},
            "layers": layers, "note": "eager launches between two events each, behind a 150 us device-side spin so that the host's launch "
                                      "preparation is not inside the figure; graph replay of the whole step is what `value` measures"}


def _physical_cores():
    try:
        avail = sorted(os.sched_getaffinity(0))
    except AttributeError:
        avail = list(range(os.cpu_count() or 1))
    seen = set()
    for c in avail:
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as fh:
                seen.add(fh.read().strip())
        except OSError:
            seen.add(str(c))
    return len(avail), max(1, len(seen))


def cpu_baseline(torch, sd, seconds):
    """The oracle (CPU restatement of the reference path, oracle/popcorn_oracle.py) timed on the host cores on bounded
    samples of the same workload: train steps (fwd + loss + bwd + clip + Adam) at B = 8 and B = 64
    on the physical cores, plus n = 8 threads (the survey container's core count) and n = 32 (where oneDNN scales best on
    these small convolutions).  `value` is the BEST leg -- the most favourable figure for the CPU."""
    from oracle import popcorn_oracle as O
    from popcorn_amd.data.synthetic import make_raw_batch
    avail, phys = _physical_cores()

    def leg(B, threads):
        torch.set_num_threads(threads)
        batch = make_raw_batch(B, 100, 100, seed=1600)
        sample = {"input": O.select_normalize(batch["raw"]), "admin_mask": batch["admin_mask"],
                  "census_idx": batch["census_idx"], "y": batch["y"]}
        params, state = dict(sd), {}

        def one():
            loss, out, grads, _ = O.train_step_grads(params, dict(sample))
            _, clipped = O.clip_grad_norm(grads, 0.01)
            params.update(O.adam_step(params, clipped, state, lr=1e-4, weight_decay=1e-5))
        t0 = time.perf_counter()
        one()                                   # warm-up (also sizes the bounded sample)
        t_warm = time.perf_counter() - t0
        n = max(1, min(200, int(seconds / max(t_warm, 1e-3))))
        t0 = time.perf_counter()
        for _ in range(n):
            one()
        dt = time.perf_counter() - t0
        return {"batch": B, "threads": threads, "steps": n, "seconds": round(dt, 2), "patches_per_s": round(B * n / dt, 2)}

    plan = [(8, phys), (64, phys), (8, 8), (8, min(32, avail))]
    legs, seen = [], set()
    for B, th in plan:
        if (B, th) in seen:
            continue
        seen.add((B, th))
        legs.append(leg(B, th))
    best = max(legs, key=lambda l: l["patches_per_s"])
    return {"value": best["patches_per_s"], "unit": "patches/s", "cores": best["threads"], "kind": "port",
            "sample": f"{best['steps']} train steps x B={best['batch']} synthetic 100x100 tiles in {best['seconds']} s (oracle = "
                      f"torch-CPU fp32 restatement; {avail} logical / {phys} physical cores available; best of the legs below)",
            "legs": legs}


def _rocprof_avg_us(kernel, precision):
    """Average duration (us) of `kernel` (FULL name incl. template arguments, e.g. ``head_bwd_pc_kernel<0>``) in the NEWEST tracked
    rocprofv3 summary of this precision (profiles/r*_<prec>_graph_kernel_stats.csv: `rocprofv3 --kernel-trace --stats` of this very bench
    command under graph replay), and the file it came from -- but ONLY when that round's stamp (profiles/r*_stamp.json, sha256 of the
    kernel sources the profile was collected from, tools/csrc_hash.py) equals the tree's: a static figure from another build would sit
    on the live line as if it belonged to it.  Returns (us, file, stamp_matches)."""
    import csv
    import glob
    import re
    rnd = lambda f: int(re.search(r"r(\d+)_", os.path.basename(f)).group(1))  # noqa: E731
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{precision}_graph_kernel_stats.csv")), key=rnd)
    if not files:
        return None, None, False
    f = files[-1]
    match = False
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from csrc_hash import csrc_hash
        stamp = json.load(open(os.path.join(ROOT, "profiles", f"r{rnd(f)}_stamp.json")))
        match = stamp.get("csrc_sha256") == csrc_hash(ROOT)[0]
    except Exception:
        match = False
    if not match:
        return None, os.path.basename(f), False
    strip = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()  # noqa: E731
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if strip(row["Name"]) == kernel:
                return round(float(row["AverageNs"]) / 1e3, 2), os.path.basename(f), True
    return None, os.path.basename(f), True


def fwd_parity(torch, margs, sd_cpu, batch, dev):
    """BASELINE's metric, second half ("fwd max-abs-err vs CPU"): the drop-in forward of config[1] -- B tiles of 100 x 100, sparse
    head over the census regions -- through the HIP path against the CPU oracle on the same tiles, same parameters."""
    from oracle import popcorn_oracle as O
    from popcorn_amd.model import get_model_kwargs, model_dict
    torch.manual_seed(1600)
    model = model_dict["POPCORN"](**get_model_kwargs(margs, "POPCORN")).to(dev).eval()
    model.load_state_dict(sd_cpu)
    x = O.select_normalize(batch["raw"].cpu())
    cpu = {"input": x, "admin_mask": batch["admin_mask"].cpu(), "census_idx": batch["census_idx"].cpu()}
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    with torch.no_grad():
        torch.manual_seed(11)
        ref = O.popcorn_forward(sd_cpu, dict(cpu), padding=False, sparse=True)
        torch.manual_seed(11)
        out = model({k: v.to(dev) for k, v in cpu.items()}, padding=False, sparse=True)
    res = {"config": f"B={x.shape[0]} 100x100 vs CPU oracle (drop-in forward, padding=False, sparse=True)", "tolerance": 1e-4}
    worst_abs = worst_rel = 0.0
    for key in ("popdensemap", "popcount", "scale"):
        a, r = out[key].float().cpu(), ref[key].float()
        if a.shape != r.shape:
            raise SystemExit(f"fwd_parity: {key} shape {tuple(a.shape)} vs oracle {tuple(r.shape)}")
        ea = (a - r).abs().max().item()
        er = ea / max(r.abs().max().item(), 1e-30)
        res[key] = {"max_abs": ea, "max_rel": er}
        worst_abs, worst_rel = max(worst_abs, ea), max(worst_rel, er)
    res["max_abs"], res["max_rel"] = worst_abs, worst_rel
    res["ok"] = worst_rel <= 1e-4
    return res


def grad_parity(torch, margs, sd_cpu, batch, dev, ntiles=8):
    """The training half of the parity statement on the bench line: ONE fused train step (eager, fp32) on the first `ntiles` tiles of the
    synthetic batch, its 56 gradients against the CPU oracle -- fp32 as it is, and fp64 under SHARED decisions (the oracle takes the HIP
    forward's side at every ReLU mask / pooling arg-max, tests/tie_adjudication.py: no tie is left to flip, the residual is rounding)."""
    try:
        from oracle import popcorn_oracle as O
        from popcorn_amd import ops
        from popcorn_amd.data import stats
        from popcorn_amd.model import get_model_kwargs, model_dict
        from popcorn_amd.train import FusedTrainStep
        from tests.tie_adjudication import forced_decision_distance, rel
        torch.manual_seed(1600)
        model = model_dict["POPCORN"](**get_model_kwargs(margs, "POPCORN")).to(dev)
        model.load_state_dict(sd_cpu)
        raw = batch["raw"][:ntiles]
        x = ops.select_normalize(raw.to(dev), stats.BAND6, stats.MEAN6, stats.STD6)
        cpu = {"input": O.select_normalize(raw.cpu()), "admin_mask": batch["admin_mask"][:ntiles].cpu(),
               "census_idx": batch["census_idx"][:ntiles].cpu(), "y": batch["y"][:ntiles].cpu()}
        tr = FusedTrainStep(model, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01)
        torch.manual_seed(13)
        loss = tr.step({"input": x, "admin_mask": cpu["admin_mask"].to(dev), "census_idx": cpu["census_idx"].to(dev), "y": cpu["y"].to(dev)})
        torch.cuda.synchronize()
        hip = {n: g.cpu() for n, g in tr.grads.items()}
        torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
        torch.manual_seed(13)
        l32, _, g32, _ = O.train_step_grads(sd_cpu, dict(cpu))
        wf, name, flips, l64 = forced_decision_distance(sd_cpu, cpu, x, hip, 13)
        return {"config": f"B={ntiles} 100x100, one fused train step vs the CPU oracle, 56 gradients, max over tensors of max-abs-err / max(|ref|, 1e-3)",
                "vs_fp32_oracle": max(rel(hip[n], g32[n]) for n in g32), "loss_rel_err": abs(loss[0].item() - l64.item()) / max(1.0, abs(l64.item())),
                "vs_fp64_oracle_under_shared_decisions": wf, "worst_tensor": name,
                "decisions_the_fp64_oracle_alone_takes_differently": {k: (len(v) if isinstance(v, list) else v) for k, v in flips.items()},
                "tolerance": 1e-4, "ok": bool(wf <= 1e-4)}
    except Exception as e:            # a checker's failure must not take the measured line with it
        return {"error": f"{type(e).__name__}: {e}"}


FLOP_FWD_PX = 2 * (9040 + 9072 + 9344)        # SURVEY.md 8d per pixel: trainable U-Net + building extractor + head = 54,912 flop


def config5_leg(torch, margs, dev, hw=(4096, 5888), passes=3):
    """BASELINE config 5 on one GPU: sliding 2048 x 2048 windows (overlap 128) over a synthetic raster, ensemble of one, device-resident
    stitcher (eval.evaluate_raster): windows/s, Mpx/s through the network, fraction of the fp32-matrix peak at 54.9 kflop / px."""
    from popcorn_amd.eval import evaluate_raster, get_patch_indices
    from popcorn_amd.model import get_model_kwargs, model_dict
    torch.manual_seed(1600)
    model = model_dict["POPCORN"](**get_model_kwargs(margs, "POPCORN")).to(dev).eval()
    raster = torch.randn(1, 6, hw[0], hw[1], device=dev)
    nwin = get_patch_indices(hw[0], hw[1]).shape[0]
    times = []
    for _ in range(passes + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mean, _, _, _ = evaluate_raster([model], raster)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    dt = statistics.median(times[1:])
    px = nwin * 2048 * 2048
    ok = bool(torch.isfinite(mean).all())
    del raster, mean
    torch.cuda.empty_cache()
    return {"workload": f"{nwin} windows of 2048x2048 (overlap 128) over a {hw[0]}x{hw[1]} synthetic raster, 1 member, fp32, stitched on the device",
            "windows_per_s": round(nwin / dt, 1), "Mpx_per_s": round(px / dt / 1e6, 1), "ms_per_window": round(dt / nwin * 1e3, 3),
            "alg_flop_per_px": FLOP_FWD_PX, "tflops": round(px * FLOP_FWD_PX / dt / 1e12, 2),
            "frac_of_fp32_mfma_peak": round(px * FLOP_FWD_PX / dt / FP32_MATRIX_PEAK, 4), "finite": ok}


def _region_alg_flop(B, H, W, nsel, enc_ng, unet_ng):
    """Algorithmic flop of one train step on B regions of H x W (SURVEY.md 8d per-pixel figures: MACs per pixel of the conv domain --
    encoder 3,888, decoder 5,152 (both streams), building extractor 9,072 on ITS padded domain, head 9,344 per selected pixel)."""
    from popcorn_amd.model.popcorn import pad_geometry
    pt, pb, pl, pr = pad_geometry(H, W, False)
    pu = B * (H + pt + pb) * (W + pl + pr)                       # the trainable U-Net's conv domain
    pbe = B * (H + 28) * (W + 28)                                # the frozen extractor pads by 14 on every side (popcorn.py:231-258)
    enc, dec, be, head = 3888, 5152, 9072, 9344
    mac = pu * (enc + dec) + pbe * be + nsel * head              # forward
    mac += 2 * nsel * head                                       # head backward (data + weight gradients)
    if not unet_ng:
        mac += pu * 2 * dec                                      # decoder: data + weight gradients
        if not enc_ng:
            mac += pu * (2 * enc - 432)                          # encoder: both, minus the data gradient of the two input convs
        else:
            mac -= pu * (576 + 288 + 64)                         # no data gradient into the skip tensors / the 32x32-level output
    return 2.0 * mac


def config3_regions_leg(torch, margs, dev, steps=10, blocks=5):
    """BASELINE config 3 as the reference RUNS it (run_train.py:186-202): weak_batch_size = 2 census regions of variable size, the
    truncation regime of each batch decided by the real limit1 / limit2 / limit3 defaults, eager steps (no graph for varying shapes) through
    the native executor (pc_train_step: one C-ABI call per step).  A seeded list of batch geometries spanning 1e5 .. 9e6 px; per shape
    `blocks` timed blocks of `steps` steps each (3 warm-up steps), the MEDIAN block is reported; Mpx/s over the whole list, algorithmic TFLOP/s."""
    from popcorn_amd import ops
    from popcorn_amd.cli import limit_regime, train_parser
    from popcorn_amd.data import stats
    from popcorn_amd.data.synthetic import make_raw_batch
    from popcorn_amd.model import get_model_kwargs, model_dict
    from popcorn_amd.train import FusedTrainStep
    a = train_parser().parse_args([])
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))      # host glue = tiny CPU ops (the trainer's setting, cli.py); the CPU legs left it wide
    torch.manual_seed(1600)
    model = model_dict["POPCORN"](**get_model_kwargs(margs, "POPCORN")).to(dev)
    tr = FusedTrainStep(model, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01, reducer=_LocalReducer())
    shapes = [(230, 220), (517, 389), (700, 640), (1030, 770), (1500, 1400), (2000, 1700), (2100, 2140), (2100, 2150)]
    rows, tot_px, tot_t, tot_fl = [], 0, 0.0, 0.0
    torch.cuda.reset_peak_memory_stats()
    for H, W in shapes:
        B = a.weak_batch_size
        enc_ng, unet_ng, skip = limit_regime(B * H * W, a.limit1, a.limit2, a.limit3)
        batch = make_raw_batch(B, H, W, seed=H * 1000 + W, region="disc")
        x = ops.select_normalize(batch["raw"].to(dev), stats.BAND6, stats.MEAN6, stats.STD6)
        smp = {"input": x, "admin_mask": batch["admin_mask"].to(dev), "census_idx": batch["census_idx"].to(dev), "y": batch["y"].to(dev)}
        torch.manual_seed(3)
        for _ in range(3):
            tr.step(dict(smp), encoder_no_grad=enc_ng, unet_no_grad=unet_ng)      # warm-up (arena growth, attribute queries)
        torch.cuda.synchronize()
        nsel = int(tr.last["mask"].sum().item())
        nb = blocks if B * H * W < 4e6 else 3
        block_ms = []
        for _ in range(nb):
            t0 = time.perf_counter()
            for _ in range(steps):
                loss = tr.step(dict(smp), encoder_no_grad=enc_ng, unet_no_grad=unet_ng)
            torch.cuda.synchronize()
            block_ms.append((time.perf_counter() - t0) / steps * 1e3)
        dt = statistics.median(block_ms) * 1e-3
        # the HOST's own cost of a step: 7 steps enqueued into an idle queue (inside the trainer's 8-step run-ahead window), best of 3 --
        # the blocks above are paced by the device as long as this is the smaller figure
        host_ms = None
        if B * H * W < 1e6:
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(7):
                    tr.step(dict(smp), encoder_no_grad=enc_ng, unet_no_grad=unet_ng)
                h = (time.perf_counter() - t0) / 7 * 1e3
                host_ms = h if host_ms is None or h < host_ms else host_ms
            torch.cuda.synchronize()
        lv = float(loss[0].item())
        if not (lv == lv) or abs(lv) == float("inf"):
            raise SystemExit(f"config3_regions: non-finite loss at {B}x{H}x{W}")
        fl = _region_alg_flop(B, H, W, nsel, enc_ng, unet_ng)
        rows.append({"batch": f"{B}x{H}x{W}", "Mpx": round(B * H * W / 1e6, 3), "regime": "head only" if unet_ng else ("decoder + head" if enc_ng else "all"),
                     "selected_px": nsel, "ms_per_step": round(dt * 1e3, 3), "Mpx_per_s": round(B * H * W / dt / 1e6, 1),
                     "tflops": round(fl / dt / 1e12, 2), "frac_of_fp32_mfma_peak": round(fl / dt / FP32_MATRIX_PEAK, 4),
                     "ms_per_step_blocks": [round(v, 3) for v in block_ms],
                     "host_enqueue_ms_per_step": None if host_ms is None else round(host_ms, 3)})
        tot_px += B * H * W; tot_t += dt; tot_fl += fl
        del smp, x, batch
        torch.cuda.empty_cache()
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    native = tr.native_steps
    del tr, model
    torch.cuda.empty_cache()
    return {"workload": "run_train.py geometry: weak_batch_size=2 census regions, sizes 1e5..9e6 px, regime per batch from the default "
                        "limit1/2/3 = 9e6/9e6/13e6, eager fused step (fp32) through the native executor (one pc_train_step call per step), "
                        "disc-shaped regions inside the crop; median of timed 10-step blocks per shape",
            "native_executor_steps": native, "steps_per_block": steps,
            "Mpx_per_s": round(tot_px / tot_t / 1e6, 1), "step_tflops": round(tot_fl / tot_t / 1e12, 2),
            "frac_of_fp32_mfma_peak": round(tot_fl / tot_t / FP32_MATRIX_PEAK, 4), "peak_hbm_gib": round(peak, 2), "batches": rows}


def small_batch_leg(torch, margs, dev, B=16, steps=60, blocks=5):
    """The headline step at B = 16 tiles (captured graph, raw 15-band tiles resident): the batch size at which per-launch fixed costs weigh
    four times as much as at B = 64 (reported on the driver's line instead of only in DESIGN.md)."""
    from popcorn_amd.data.synthetic import make_raw_batch
    from popcorn_amd.model import get_model_kwargs, model_dict
    from popcorn_amd.train import FusedTrainStep
    torch.manual_seed(1600)
    model = model_dict["POPCORN"](**get_model_kwargs(margs, "POPCORN")).to(dev)
    tr = FusedTrainStep(model, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01, loss=("log_l1_loss",), lam=(1.0,), scale_regularization=0.01,
                        lam_weak=100.0, reducer=_LocalReducer(), use_graph=True)
    batch = make_raw_batch(B, 100, 100, seed=1616, device=dev)
    smp = tr.static_buffers(B, 100, 100, raw_channels=batch["raw"].shape[1])
    smp["raw"].copy_(batch["raw"]); smp["admin_mask"].copy_(batch["admin_mask"]); smp["census_idx"].copy_(batch["census_idx"]); smp["y"].copy_(batch["y"])
    torch.manual_seed(1616)
    for _ in range(200):
        tr.step(smp)
    torch.cuda.synchronize()
    ms = []
    for _ in range(blocks):
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = tr.step(smp)
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t0) / steps * 1e3)
    lv = float(loss[0].item())
    if not (lv == lv) or abs(lv) == float("inf"):
        raise SystemExit("small_batch: non-finite loss")
    m = statistics.median(ms)
    del tr, model
    torch.cuda.empty_cache()
    return {"batch": B, "value": round(B / m * 1e3, 1), "unit": "patches/s", "ms_per_step": round(m, 4), "ms_per_step_blocks": [round(v, 4) for v in ms],
            "step_frac_of_fp32_mfma_peak": round(B / m * 1e3 * FLOP_TRAIN_PER_TILE / FP32_MATRIX_PEAK, 4),
            "workload": f"the headline step at B = {B} tiles (captured graph, raw 15-band tiles resident, every pixel selected)"}


def config3_epoch_leg(torch, dev, regions=256, hw=(150, 700), workers=0):
    """BASELINE config 3 END TO END: the trainer counterpart's own loop (popcorn_amd.cli.Trainer.train = run_train.py:146-269) over
    `regions` synthetic census regions through the DataLoader -- collate (zero-padding to the batch maximum), pinned staging, the
    one-batch-ahead copy stream (data/feed.py: RegionFeed), the one-launch augmentation (pc_augment_raw, coins drawn with the reference's
    generators), the native executor with the normalisation inside its ingest -- timed over a whole epoch (the epoch before it is untimed:
    worker start-up, per-worker sample caches, arena growth), next to the SAME batches already prepared and resident on the device."""
    import tempfile
    from popcorn_amd.cli import Trainer, limit_regime, prepare_sample_fused, train_parser
    from popcorn_amd.data.feed import RegionFeed
    tmp = tempfile.mkdtemp(prefix="pc_epoch_")
    argv = (f"-S2 -NIR -S1 -occmodel -senbuilds -pret -wd 1e-5 --biasinit 0.9407 -lr 1e-4 --synthetic_regions {regions} -wb 2 --save_dir {tmp} "
            f"-lt 1000000 -val 1000000 -e 1 --synthetic_hw_range {hw[0]} {hw[1]} --save-model no -w {workers}").split()
    import contextlib
    t = Trainer(train_parser().parse_args(argv))
    a = t.args
    with contextlib.redirect_stdout(sys.stderr):        # (the trainer's "Training finished" line: stdout carries the ONE JSON line)
        t.train()                               # epoch 0: untimed
        torch.cuda.synchronize()
        a.num_epochs = 2
        native0, it0 = t.fused.native_steps, t.info["iter"]
        t0 = time.perf_counter()
        t.train()                               # epoch 1: timed, loader -> collate -> feed -> augment -> step
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    steps = t.info["iter"] - it0
    # the same epoch's batches, prepared and resident: what the executor alone takes on this geometry mix
    staged, px = [], 0
    for smp in RegionFeed(t.loader, dev):
        s = prepare_sample_fused(smp, t.data_transform)
        n = s["raw"].shape[0] * s["raw"].shape[2] * s["raw"].shape[3]
        staged.append((s, limit_regime(n, a.limit1, a.limit2, a.limit3)))
        px += n
    for s, (e, u, k) in staged[:3]:
        t.fused.step(s, encoder_no_grad=e, unet_no_grad=u)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for s, (e, u, k) in staged:
        if not k:
            t.fused.step(s, encoder_no_grad=e, unet_no_grad=u)
    torch.cuda.synchronize()
    dr = time.perf_counter() - t1
    loss = float(t.fused.loss_out[0].item())
    if not (loss == loss) or abs(loss) == float("inf"):
        raise SystemExit("config3_epoch: non-finite loss")
    res = {"workload": f"Trainer.train() (run_train.py loop counterpart) over {regions} synthetic census regions of {hw[0]}..{hw[1]} px sides, "
                       f"weak_batch_size 2, augmentations on, DataLoader (num_workers = {workers}) pulled by the feed's producer thread, pinned staging ring, copy "
                       "stream, one-launch augmentation, native executor (raw input form: normalisation inside the step); second epoch timed",
           "steps": steps, "native_executor_steps": t.fused.native_steps - native0, "regions_per_s": round(2 * steps / dt, 1),
           "Mpx_per_s": round(px / dt / 1e6, 1), "ms_per_step": round(dt / max(steps, 1) * 1e3, 3),
           "resident": {"ms_per_step": round(dr / max(len(staged), 1) * 1e3, 3), "Mpx_per_s": round(px / dr / 1e6, 1)},
           "ratio_to_resident": round(dr / dt, 3)}
    if getattr(t.loader, "_iterator", None) is not None:
        t.loader._iterator._shutdown_workers()
    del t, staged
    torch.cuda.empty_cache()
    return res


def h2d_leg(torch, trainer, sample, batch, band_sel, nsteps, label):
    """The train step with every batch coming from PINNED HOST memory.  The loader owns TWO static sets of the trainer
    (`static_buffers(slot=0 / 1)`: raw tile + one packed buffer {admin_mask, y, census_idx}), each captured into its own graph: a copy
    stream moves batch i+1 straight into the idle set (2 H2D copies, no staging copy, no kernel outside the graph) while the graph of
    the other set computes step i; two events per set order the two streams.  The reference's loop does the same H2D every step
    (run_train.py:186, utils/utils.py:22-27), synchronously."""
    from popcorn_amd.data import stats
    B, _, H, W = batch["raw"].shape
    keep_norm = trainer.raw_norm
    packed = trainer.pack_small(batch["admin_mask"].cpu(), batch["y"].cpu(), batch["census_idx"].cpu()).pin_memory()
    from popcorn_amd.data.feed import HostFeed
    if band_sel == "split":
        # what a loader reads from disk: the 4 selected S2 bands as the GeoTIFF's uint16 digital numbers + the 2 S1 bands as fp32
        b6 = list(stats.BAND6)
        trainer.raw_norm = (tuple(range(6)), stats.MEAN6, stats.STD6)
        host = {"_rawpacked": trainer.pack_split(batch["raw"][:, b6[:4]].round().to(torch.int32).cpu().to(torch.uint16).contiguous(),
                                                 batch["raw"][:, b6[4:]].contiguous().cpu()).pin_memory(), "_packed": packed}
        feed = HostFeed(trainer, B, H, W, kind="split")
    else:
        raw = batch["raw"] if band_sel is None else batch["raw"][:, list(band_sel)].contiguous()
        if band_sel is not None:             # the host already holds the 6 model bands: the ingest kernel only normalises + pads
            trainer.raw_norm = (tuple(range(6)), stats.MEAN6, stats.STD6)
        host = {"raw": raw.cpu().pin_memory(), "_packed": packed}
        feed = HostFeed(trainer, B, H, W, kind="raw", raw_channels=raw.shape[1])

    def run(n):
        # (popcorn_amd/data/feed.py: batch i + 1 is copied into the idle set on a side stream -- one whose hardware queue is MEASURED not to
        # be the compute stream's -- while the graph of the other set computes step i)
        for _ in range(n):
            feed.step(host)
        feed.flush()
    # untimed block first: graph captures of the two sets + ~0.5 s at this leg's own load (the FIRST leg after the bf16 section read 5 %
    # slow whatever its feed -- tools/h2d_legs.py runs the legs in both orders: the first one of a sequence is the slow one, resident
    # control included -- the clock / power state of the previous section, not the feed)
    run(max(8, 300))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(nsteps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    trainer.raw_norm = keep_norm
    nbytes = sum(v.numel() * v.element_size() for v in host.values())
    return {"feed": label, "steps": nsteps, "ms_per_step": round(dt / nsteps * 1e3, 4), "host_bytes_per_step": nbytes,
            "h2d_gbps_sustained": round(nbytes * nsteps / dt / 1e9, 2), "copy_stream": hex(feed.cs.cuda_stream)}, dt


def extra_legs(torch, dist, args, world, rank, dev, B, batch, trainer, sample, step, build, run_blocks, timed_block):
    """`soak`, `h2d`, `bf16` (see the module docstring).  Every leg runs the same code on every rank (barrier-bracketed
    like the headline), so the multi-GPU line carries them too; `soak` only at N = 1."""
    out = {}
    sel6 = None

    def agg(dt):            # max over ranks of a wall time
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return t.item()
        return dt

    # ---- bf16 (BASELINE config 4's precision) in the same process; skipped when the headline itself was asked in bf16
    if args.precision == "fp32":
        model16, tr16, smp16, step16, _ = build("bf16")
        torch.manual_seed(1600 + rank)
        for _ in range(max(args.warmup, 1)):
            step16()
        blocks, _, loss16 = run_blocks(step16, args.steps, args.repeats)
        dt = statistics.median(blocks)
        v16 = B * world * args.steps / dt
        o = {"value": round(v16, 1), "unit": "patches/s", "ms_per_step": round(dt / args.steps * 1e3, 4),
             "ms_per_step_blocks": [round(b / args.steps * 1e3, 4) for b in blocks], "final_loss": round(loss16, 6),
             "dtype": "bf16 (MFMA operands + stored activations; fp32 accumulate, master weights, Adam)",
             "step_frac_of_bf16_mfma_peak": round(v16 * FLOP_TRAIN_PER_TILE / (BF16_MATRIX_PEAK * world), 4)}
        if world == 1 and rank == 0 and not args.no_class_sweep:
            from popcorn_amd import _lib as L
            with L.precision("bf16"):
                sw = conv_class_sweep(torch, tr16, smp16)
            o["roofline"] = {"bound": "hbm", "kernel": "conv class of the bf16 step (every conv / transposed-conv launch, channels-last bf16)",
                             "achieved": sw["hbm_view"]["achieved_gbps"], "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                             "frac": sw["hbm_view"]["frac"], "launches": sw["launches"], "us": sw["us"],
                             "alg_mbytes": sw["alg_mbytes"], "mfma_view": sw["mfma_view"], "by_kind": sw["by_kind"]}
        out["bf16"] = o
    # ---- h2d: the headline precision, fed from pinned host memory
    from popcorn_amd.data import stats
    legs = []
    for band_sel, label in ((stats.BAND6, "6 pre-selected bands, fp32 (240 KB / tile)"),
                            ("split", "6 pre-selected bands, S2 as uint16 digital numbers + S1 fp32 (160 KB / tile)"),
                            (None, "15-band tile (600 KB / tile), band select on the device")):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        nfeed = max(args.steps * 4, 100)
        leg, dt = h2d_leg(torch, trainer, sample, batch, band_sel, nfeed, label)
        dt = agg(dt)
        leg["ms_per_step"] = round(dt / nfeed * 1e3, 4)
        leg["value"] = round(B * world * nfeed / dt, 1)
        legs.append(leg)
    from popcorn_amd import ops as _ops

    def resident_step():
        return step()
    sample["admin_mask"].copy_(batch["admin_mask"]); sample["census_idx"].copy_(batch["census_idx"]); sample["y"].copy_(batch["y"])
    dt_res, _, _ = timed_block(resident_step, nfeed)
    # what feeding costs at scale: bytes per step of the DEFAULT feed (the 6 selected bands -- what the reference's loader ships,
    # data/PopulationDataset.py:566-568 -- fp32) at the resident step rate, per rank and for 8 ranks through one host, next to what one
    # pinned hipMemcpyAsync stream sustains on this host (256 MB, median of 5)
    probe = torch.empty(256 * 2 ** 20, dtype=torch.uint8).pin_memory()
    dprobe = torch.empty_like(probe, device=dev)
    cp = []
    for _ in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dprobe.copy_(probe, non_blocking=True)
        torch.cuda.synchronize()
        cp.append(probe.numel() / (time.perf_counter() - t0) / 1e9)
    pinned_gbps = statistics.median(cp[1:])
    del probe, dprobe
    need = legs[0]["host_bytes_per_step"] / (dt_res / nfeed) / 1e9
    out["h2d"] = {"unit": "patches/s", "precision": args.precision, "legs": legs, "default_feed": legs[0]["feed"],
                  "h2d_gbps_needed_per_rank": round(need, 2), "h2d_gbps_needed_8_ranks": round(8 * need, 1),
                  "host_pinned_gbps_measured": round(pinned_gbps, 1),
                  "resident_same_block": {"steps": nfeed, "ms_per_step": round(dt_res / nfeed * 1e3, 4),
                                          "value": round(B * world * nfeed / dt_res, 1)},
                  "note": "whole job, max over ranks; pinned host -> the idle one of two static input sets (one captured graph each) on a "
                          "copy stream, compute waits on the copy event; the headline `value` keeps its inputs resident in HBM"}
    # restore the resident inputs of the static sample
    sample["admin_mask"].copy_(batch["admin_mask"]); sample["census_idx"].copy_(batch["census_idx"]); sample["y"].copy_(batch["y"])

    # ---- soak: one long block (N = 1)
    if world == 1 and args.soak_steps > 0:
        from popcorn_amd import ops

        ps = PowerSampler(torch)
        ps.start()
        dt, _, _ = timed_block(resident_step, args.soak_steps)
        pw = ps.stop()
        out["soak"] = {"steps": args.soak_steps, "ms_per_step": round(dt / args.soak_steps * 1e3, 4),
                       "value": round(B * args.soak_steps / dt, 1), "unit": "patches/s", "precision": args.precision,
                       "board_power_w": pw["board_power_w"], "sclk_mhz": pw["sclk_mhz"], "sclk_mhz_min": pw["sclk_mhz_min"],
                       "power_note": "hwmon power1 / freq1 of this device sampled every 20 ms during the block (None: sysfs not readable)"}

    return out


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None:
        sys.exit(relaunch_distributed(args))             # child launcher; nothing has touched the GPU in this process
    import torch
    import torch.distributed as dist
    from popcorn_amd.distributed import FlatReducer, init_from_env
    from popcorn_amd.data import stats
    from popcorn_amd.data.synthetic import make_raw_batch
    from popcorn_amd import ops
    from popcorn_amd.model import Args, get_model_kwargs, model_dict
    from popcorn_amd.train import FusedTrainStep

    rank, local_rank, world = init_from_env()
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a number for the wrong world size",
                  file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path)")
    ndev = torch.cuda.device_count()
    if world > ndev and not os.environ.get("POPCORN_DIST_BACKEND"):
        raise SystemExit(f"bench.py: {world} ranks but {ndev} GPUs (set POPCORN_DIST_BACKEND=gloo for a functional shared-GPU run)")
    local_rank %= max(1, ndev)                           # (functional gloo runs with more ranks than GPUs share a device)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    B = args.batch

    # rwa recipe (README.md:187): -S2 -NIR -S1 -occmodel -senbuilds -pret -wd 1e-5 --biasinit 0.9407
    margs = Args(Sentinel1=True, NIR=True, Sentinel2=True, feature_extractor="DDA", occupancymodel=True, pretrained=True,
                 biasinit=0.9407, sentinelbuildings=True)
    batch = make_raw_batch(B, 100, 100, seed=1600 + rank, device=dev)           # resident in HBM before timing

    def build(precision):
        """model + fused trainer + the static sample the captured graph reads + the step closure, for one precision"""
        torch.manual_seed(1600)
        model = model_dict["POPCORN"](**get_model_kwargs(margs, "POPCORN")).to(dev)
        sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        model.set_precision(precision)
        tr = FusedTrainStep(model, lr=1e-4, weight_decay=1e-5, gradient_clip=0.01, loss=("log_l1_loss",), lam=(1.0,),
                            scale_regularization=0.01, lam_weak=100.0, reducer=FlatReducer(), use_graph=not args.no_graph)
        # the loader side of the step writes into the tensors the captured graph reads (no per-step input copies)
        smp = tr.static_buffers(B, 100, 100) if not args.no_graph else \
            {"input": torch.empty(B, 6, 100, 100, device=dev), "admin_mask": torch.empty(B, 100, 100, device=dev),
             "census_idx": torch.empty(B, dtype=torch.int64, device=dev), "y": torch.empty(B, device=dev)}
        smp["admin_mask"].copy_(batch["admin_mask"])
        smp["census_idx"].copy_(batch["census_idx"])
        smp["y"].copy_(batch["y"])
        ops.select_normalize(batch["raw"], stats.BAND6, stats.MEAN6, stats.STD6, out=smp["input"])
        # the step proper starts from the RAW 15-band tile: band select + normalise (+ the reflect padding in fp32 mode) are its
        # first launch, inside the replayed graph (FusedTrainStep: sample key "raw"); `smp` (normalised "input") serves the
        # per-kernel roofline sections and the h2d legs
        if args.no_graph:
            raw_smp = {"raw": batch["raw"], "admin_mask": smp["admin_mask"], "census_idx": smp["census_idx"], "y": smp["y"]}
        else:
            raw_smp = tr.static_buffers(B, 100, 100, raw_channels=batch["raw"].shape[1])
            raw_smp["raw"].copy_(batch["raw"])
            raw_smp["_packed"].copy_(smp["_packed"])

        def one_step():
            return tr.step(raw_smp)
        one_step.sample = raw_smp
        return model, tr, smp, one_step, sd

    model, trainer, sample, step, sd_cpu = build(args.precision)

    torch.manual_seed(1600 + rank)
    prewarm_steps = 0
    if args.prewarm_seconds > 0:
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        while True:                                    # same count on every rank: the decision is rank 0's
            for _ in range(50):
                step()
            torch.cuda.synchronize()
            prewarm_steps += 50
            go = torch.tensor([1.0 if time.perf_counter() - t0 < args.prewarm_seconds else 0.0], device=dev)
            if world > 1:
                dist.broadcast(go, 0)
            if go.item() == 0.0:
                break
    for _ in range(max(args.warmup, 1)):
        step()

    def timed_block(step, nsteps):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            loss = step()
        torch.cuda.synchronize()
        own = time.perf_counter() - t0              # this rank's own time (before the closing barrier)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        return dt, loss, own

    def run_blocks(step, nsteps, repeats):
        blocks, owns, loss = [], [], None
        for _ in range(max(1, repeats)):
            dt, loss, own = timed_block(step, nsteps)
            blocks.append(dt)
            owns.append(own)
        loss_val = float(loss[0].item())
        if not (loss_val == loss_val) or abs(loss_val) == float("inf"):
            raise SystemExit(f"non-finite loss {loss_val}")
        return blocks, owns, loss_val

    blocks, owns, loss_val = run_blocks(step, args.steps, args.repeats)
    dt = statistics.median(blocks)
    own_ms = statistics.median(owns) / args.steps * 1e3
    if world > 1:
        t = torch.zeros(world, device=dev, dtype=torch.float64)
        t[rank] = own_ms
        dist.all_reduce(t)
        per_rank = [round(v, 4) for v in t.tolist()]
    else:
        per_rank = [round(own_ms, 4)]

    extras = {}
    if not args.no_extras and (world == 1 or args.extras_multi):
        extras = extra_legs(torch, dist, args, world, rank, dev, B, batch, trainer, sample, step, build, run_blocks, timed_block)
    elif not args.no_extras:
        # N > 1 (round 6): the scaling run shows the one limit SURVEY 8e predicted -- N ranks fed through one host -- by default: ONE host-feed
        # leg with the narrowest feed (S2 as uint16 digital numbers + S1 fp32, 160 KB per tile: 8 ranks need 49 GB/s of pinned H2D against
        # 52.5 measured on the one-link test box; the fp32 6-band feed needs 73.6), reported next to the resident `value` as `value_fed`
        torch.cuda.synchronize()
        dist.barrier()
        nfeed = max(args.steps * 4, 100)
        leg, dtf = h2d_leg(torch, trainer, sample, batch, "split", nfeed, "6 pre-selected bands, S2 as uint16 digital numbers + S1 fp32 (160 KB / tile)")
        tt = torch.tensor([dtf], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dtf = tt.item()
        leg["ms_per_step"] = round(dtf / nfeed * 1e3, 4)
        leg["value"] = round(B * world * nfeed / dtf, 1)
        extras = {"value_fed": leg["value"],
                  "h2d": {"unit": "patches/s", "precision": args.precision, "legs": [leg], "default_feed": leg["feed"],
                          "h2d_gbps_per_rank_sustained": leg["h2d_gbps_sustained"],
                          "note": "whole job, max over ranks: every rank feeds its own batches from pinned host memory on its measured copy stream; "
                                  "`value` (the headline) keeps its inputs resident in HBM"}}

    # which devices took part (one entry per rank: "<rank>:<device ordinal>:<bus id>"), gathered through the job's own backend
    dp = {"ranks_seen": 1, "devices": None, "capture_failed": False}
    if world > 1:
        pr = torch.cuda.get_device_properties(dev)
        mine = f"{rank}:{local_rank}:{getattr(pr, 'pci_bus_id', '?')}"
        seen = [None] * world
        dist.all_gather_object(seen, mine)
        flag = torch.tensor([1 if trainer.reducer.capture_failed else 0], device=dev, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        dp = {"ranks_seen": len(set(seen)), "devices": seen, "capture_failed": bool(flag.item())}
    if rank == 0:
        tiles = B * world * args.steps
        value = tiles / dt
        res = {
            "metric": "training patches/s (15-band 100x100)", "value": round(value, 1), "unit": "patches/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "bf16 (MFMA operands + stored activations; fp32 accumulate, master weights, Adam)",
            "data": "synthetic",
            "config": {"workload": f"config[1]/[2]: batch={B} synthetic S1+S2 100x100 tiles per GPU, full train step "
                                   "(building extractor + DDA dual-stream U-Net + sparse head fwd/bwd, log-L1 loss, "
                                   "clip 0.01, Adam, rwa flags), every pixel of every tile selected",
                       "global_batch": B * world, "tile": "15x100x100 -> 6x100x100 (128x128 internal)",
                       "parallelism": f"dp{world}", "graph": not args.no_graph,
                       # fp32 mode: the sparse head's products are exact 3-way bf16 operand splits on the bf16 matrix pipe, accumulated in
                       # fp32 (popcorn_hip.h: pc_set_head_split; fp32 accuracy measured against float64 in tests/test_gpu_convt_head.py)
                       "head_products": (("split3_bf16_fp32acc" if ops.L.lib().pc_get_head_split() else "fp32_mfma") if args.precision == "fp32"
                                         else "bf16"),
                       # ... and so do the fused conv backward (round 6: popcorn_hip.h: pc_set_conv_split; 7 of the step's 33 launches) and every
                       # 8- / 16-channel forward conv (csrc/conv3x3_fwd_s3.h; 7 launches); still v_mfma_f32_16x16x4_f32: the first layers, the
                       # 32 x 32 level, the composed Up backward
                       "conv_bwd_products": (("split3_bf16_fp32acc" if ops.L.lib().pc_get_conv_split() else "fp32_mfma") if args.precision == "fp32"
                                             else "bf16"),
                       "conv_fwd_products": (("split3_bf16_fp32acc" if ops.L.lib().pc_get_conv_split() else "fp32_mfma") if args.precision == "fp32"
                                             else "bf16"),
                       "backend": (dist.get_backend() if dist.is_initialized() else None),
                       "collectives": bool(trainer.reducer.active),
                       "collectives_per_step": 2 if trainer.reducer.active else 0,
                       "rccl_ranks_seen": dp["ranks_seen"] if (dist.is_initialized() and dist.get_backend() == "nccl") else None,
                       "ranks_seen": dp["ranks_seen"], "rank_devices": dp["devices"], "dp_capture_failed": dp["capture_failed"],
                       # how the captured step holds them: "one" = both inside the step's single HIP graph (RCCL), "split" =
                       # three graphs with the two collectives launched between them (gloo, or a failed capture)
                       "dp_graph": (None if (args.no_graph or not trainer.reducer.active or trainer._graphs is None) else
                                    ("one" if len(trainer._graphs[3]) == 1 else "split"))},
            "timing": f"median of {len(blocks)} blocks of {args.steps} steps, each bracketed by barrier + synchronize, max over ranks; "
                      f"{prewarm_steps} untimed steps ({args.prewarm_seconds} s) + {args.warmup} warm-up steps in front (clock settling, see --prewarm-seconds)",
            "prewarm_steps": prewarm_steps,
            "ms_per_step_blocks": [round(b / args.steps * 1e3, 4) for b in blocks],
            "per_rank_ms_per_step": per_rank,
            "final_loss": round(loss_val, 6),
            "step_tflops": round(value * FLOP_TRAIN_PER_TILE / 1e12, 3),
            "step_frac_of_fp32_mfma_peak": round(value * FLOP_TRAIN_PER_TILE / (FP32_MATRIX_PEAK * world), 4),
        }
        from popcorn_amd import _lib as L
        with L.precision(args.precision):
            res["roofline_head"] = head_bwd_roofline(torch, trainer, sample)
            # `roofline` = the kernel with the largest share of the step in the tracked rocprofv3 summary: the fused conv backward in the
            # fp32 step; in bf16 mode (and with the split form switched off) the head backward, as before
            res["roofline"] = fused_conv_bwd_roofline(torch, trainer, sample) or res["roofline_head"]
            res["roofline_conv"] = conv_kernel_roofline(torch, B)
        if world == 1 and not args.no_class_sweep:
            res["roofline_conv_class"] = conv_class_sweep(torch, trainer, sample)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(torch, sd_cpu, args.cpu_seconds)
        else:
            res["cpu_baseline"] = None
        res.update(extras)
        if world == 1 and not args.no_config_legs and args.precision == "fp32":
            # BASELINE's other configurations on the same line: forward parity of config[1] against the CPU oracle (the metric's
            # second half), config[2] at the reference's real region geometry, config[4]'s sliding windows
            res["fwd_parity"] = fwd_parity(torch, margs, sd_cpu, batch, dev)
            res["grad_parity"] = grad_parity(torch, margs, sd_cpu, batch, dev)
            del trainer, model
            torch.cuda.empty_cache()
            res["batch16"] = small_batch_leg(torch, margs, dev, 16)
            res["config3_regions"] = config3_regions_leg(torch, margs, dev)
            res["config3_epoch"] = config3_epoch_leg(torch, dev)
            res["config5"] = config5_leg(torch, margs, dev)
    else:
        res = None
    # RCCL writes its version banner through C stdio, which (stdout not being a terminal) would otherwise be flushed at exit,
    # i.e. BEHIND the result: every rank drains it before the last barrier, so that the JSON line is the last line of the stream
    import ctypes
    libc = ctypes.CDLL(None)
    libc.fflush(None)
    if dist.is_initialized():               # world > 1, or the forced single-rank collectives (POPCORN_DIST_FORCE=1)
        dist.barrier()
        dist.destroy_process_group()
        libc.fflush(None)
        if world > 1:
            time.sleep(0.5 if rank == 0 else 0.0)        # the other ranks' last flush
    if res is not None:
        print(json.dumps(res), flush=True)
    if dp["capture_failed"]:
        sys.exit(3)              # the line above is printed, but a data-parallel graph capture that failed is not a clean run


if __name__ == "__main__":
    main()
