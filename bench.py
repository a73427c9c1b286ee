#!/usr/bin/env python3
"""Benchmark of the VAME RNN-VAE training hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` without a torchrun environment launches the N ranks itself (re-exec under
torch.distributed.run, 127.0.0.1 rendezvous); it never falls back to fewer ranks than asked for.

A "step" is one pass of the hot path over one batch of synthetic temporal windows: window gather
(sliding-window batcher) -> RNN-VAE forward -> MSE + future-MSE + KL + nuclear-norm loss -> BPTT
-> one RCCL all-reduce of the flat gradient bucket (N > 1) -> fused Adam-AMSGrad.  Workload =
BASELINE.json configs[1]: T=30, F=24, H=256, Z=30, FS=15, batch 4096 windows per GPU, fp32.
The series is already resident in HBM when the timed region starts.

Prints ONE JSON line (rank 0): metric/value/unit + roofline (dominant kernel, measured with HIP
events on the launch stream; HBM-bound kernels of the step next to it) + cpu_baseline
(reference-equivalent torch-CPU model, oracle/torch_ref.py) + `repeat_spread` (two more timed regions
of the same length) + `also` (N = 1: BASELINE configs[3] shape, a configs[4]-shape embedding pass with its CPU baseline, the stock batch
256, each with its own roofline, and `split_gemm`: the opt-in split-bf16 contractions -- weight gradients, layer-1 projections and their data gradients) + `distributed` (N > 1: proof of the RCCL path and a same-run 1-rank leg).
"""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

F, Z, FS = 24, 30, 15
N_SERIES = 1_000_000
PEAK_F32_MFMA_TFLOPS = 157.3              # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0
MFLOP_PER_WINDOW_EMBED = 96.707           # encoder + Lambda mean, H=256, T=30


def train_mflop_per_window(H, T):
    """matmul flops, 2/MAC, bwd = 2x fwd, decoder input projection once (400.343 at H=256,T=30; 3012.526 at 512/60)."""
    fwd = 2 * T * 2 * 3 * H * (F + H) + 2 * T * 2 * 3 * H * (2 * H + H) + 2 * 2 * 4 * H * Z         # encoder L0 + L1, Lambda
    for st in (T, FS):                                                                             # decoder, future decoder
        fwd += 2 * Z * 2 * H + 2 * 2 * 3 * H * Z + 2 * st * 2 * 3 * H * H + 2 * st * 2 * H * F
    return 3 * fwd / 1e6


def synth_series(n, seed=0):
    rng = np.random.default_rng(seed)
    i = np.arange(n, dtype=np.float64)
    X = np.sin(2 * np.pi * i[None, :] * (np.arange(F, dtype=np.float64)[:, None] + 1) / 997.0) + 0.5 * rng.standard_normal((F, n))
    return ((X - X.mean()) / X.std()).astype(np.float32)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args):
    """--gpus N > 1 outside a torchrun environment: become `python -m torch.distributed.run --nproc-per-node N ... bench.py <same args>`
    (one process per GPU, RCCL).  VAME_BENCH_LAUNCHER (tests only) names a wrapper script that is run in place of bench.py and
    receives bench.py as its first argument -- the CPU test-suite passes its emulator harness there."""
    wrapper = os.environ.get("VAME_BENCH_LAUNCHER")
    script = [wrapper, os.path.abspath(__file__)] if wrapper else [os.path.abspath(__file__)]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port())] + script + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


# ------------------------------------------------------------------------------------------------ per-kernel timing (HIP events)
class KernelTimer:
    """HIP-event timing of each C-ABI launch group on the stream it is launched on (torch's current stream).  MFMA-bound groups
    are also bracketed by clock stamps (vame_clock_stamp, OUTSIDE the event pair): the average shader clock while that kernel ran."""

    def __init__(self, clocked=()):
        self.records = []
        self.clocked = set(clocked)

    def wrap(self, ops_mod, name, work_fn):
        inner = getattr(ops_mod, name)
        from vame_amd import ops as _ops

        def timed(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            probe = _ops.ClockProbe(torch.device("cuda", torch.cuda.current_device())) if name in self.clocked else None
            if probe is not None:
                probe.start()
            e0.record()
            r = inner(*a, **k)
            e1.record()
            if probe is not None:
                probe.stop()
            self.records.append((name, work_fn(*a, **k), e0, e1, probe))
            return r
        setattr(ops_mod, name, timed)
        return inner

    def summary(self):
        """Sums per kernel key.  A launch whose event pair reads more than 20x the median of its key's launches is counted at that median and
        reported in `timer_outliers` (one 3-step run on a gpurun box once put the dominant row 1000x low, not reproduced since: one bad pair is
        enough for that); nothing else is filtered."""
        torch.cuda.synchronize()
        agg, per_key = {}, {}
        timed = [(name, key, work, e0.elapsed_time(e1), probe) for name, (key, work), e0, e1, probe in self.records]
        for _, key, _, ms, _ in timed:
            per_key.setdefault(key, []).append(ms)
        med = {k: sorted(v)[len(v) // 2] for k, v in per_key.items()}
        for name, key, work, ms, probe in timed:
            d = agg.setdefault(key, dict(api=name, launches=0, ms=0.0, work=0.0, clk_ms=0.0, clk_w=0.0, outliers=0))
            if len(per_key[key]) >= 3 and ms > 20.0 * med[key]:
                d["outliers"] += 1
                ms, probe = med[key], None
            d["launches"] += 1
            d["ms"] += ms
            d["work"] += work
            mhz = probe.mhz() if probe is not None else None
            if mhz:                                   # time-weighted mean clock of the group's launches
                d["clk_ms"] += ms
                d["clk_w"] += ms * mhz
        for d in agg.values():
            d["mhz"] = d["clk_w"] / d["clk_ms"] if d["clk_ms"] > 0 else None
        return agg


def profile_kernels(model, loader, B, steps=3):
    """Per-kernel-class time + algorithmic flops (MFMA classes) or algorithmic HBM bytes (gather / mse / timesum) over `steps` train
    steps, separate from the timed region."""
    from vame_amd import ops
    mfma_apis = ("gru_seq_fwd", "gru_seq_bwd", "gru_wide_fwd", "gru_wide_bwd", "gru_coop_fwd", "gru_coop_bwd", "gemm", "gemm_group")
    kt = KernelTimer(clocked=mfma_apis)

    def gru_flops(tag):
        def f(streams, B_, Hh, *a, **k):
            key_t = ops.GF["T"] if tag == "fwd" else ops.GB["T"]
            fl = sum(2.0 * 3 * Hh * Hh * B_ * int(s[key_t]) for s in streams)
            name = "gru_seq" if Hh <= 256 else "gru_wide"
            # launches of one kernel family that stream different things are different roofline rows: the fused input projection (`xin`),
            # a per-step gi tile (`gi`), a time-constant gi (`const-gi`: the decoders); T = steps of the longest stream
            form = ""
            if tag == "fwd":
                s0 = streams[0]
                form = " xin" if s0.get(ops.GF["WPX"]) else (" gi" if int(s0[ops.GF["GI_T"]]) != 0 else " const-gi")
            else:
                form = " dy" if streams[0].get(ops.GB["DY"]) else " no-dy"      # BPTT with / without a gradient tile per step (encoder layer 1 has none)
            steps_ = max(int(s[key_t]) for s in streams)
            return (f"{name}_{tag}_kernel<{Hh}> x{len(streams)} streams{form} T={steps_}", fl)
        return f

    def coop_flops(tag):       # column-split GRU launches of the small-batch path: rows = (row0, nrows) restricts the launch to a row range
        def f(streams, B_, Hh, state, rows=(0, 0), **k):
            key_t = ops.GF["T"] if tag == "fwd" else ops.GB["T"]
            nrows = rows[1] or B_
            return (f"gru_coop_{tag}_kernel<{Hh}> x{len(streams)} streams", sum(2.0 * 3 * Hh * Hh * nrows * int(s[key_t]) for s in streams))
        return f

    # contractions with a dimension <= 32 (K = 24 / 30 projections of the features / of z, the 24- / 30-wide heads and their weight gradients)
    # stream their large operand once at HBM speed with the matrix pipes nearly idle (MFMA busy 0.05-0.31): they are rows of the HBM table
    # (algorithmic bytes = A, B read once + C written once), not of an MFMA class
    def gemm_flops(M, N, K, A, akm, Bm, bkm, *a, **k):
        kind = "NT" if (not akm and not bkm) else ("NN" if not akm else "TN")
        if min(M, N, K) <= 32:
            return (f"hbm gemm_kernel {kind} narrow (a dimension <= 32)", 4.0 * (M * K + K * N + M * N))
        sp = k.get("split")
        return (f"gemm_kernel {kind} M={M} N={N} K={K}" + ("" if sp is None else f" bf16x6 opt={sp}"), 2.0 * M * N * K)

    def group_flops(M, N, K, As, akm, Bs, bkm, *a, **k):
        kind = "NT" if (not akm and not bkm) else ("NN" if not akm else "TN")
        if min(M, N, K) <= 32:
            return (f"hbm gemm_kernel {kind} narrow (a dimension <= 32)", 4.0 * (M * K + K * N + M * N) * len(As))
        sp = k.get("split")
        return (f"gemm_kernel {kind} M={M} N={N} K={K} x{len(As)} grouped" + ("" if sp is None else f" bf16x6 opt={sp}"), 2.0 * M * N * K * len(As))

    # HBM-bound kernels: algorithmic bytes = every element read once + every element written once
    def gather_bytes(X, N, F_, starts, start0, B_, L, out):
        return ("hbm window_gather_kernel", 2.0 * 4 * B_ * L * F_)

    def mse_bytes(pred, target, tgt_off, tgt_row, B_, TF, *a, **k):
        return ("hbm mse_kernel", 3.0 * 4 * B_ * TF)                       # prediction + target in, dpred out

    def head_bytes(Yop, M, F_, K, *a, **k):
        return ("hbm head_stream_kernel", 4.0 * (2.0 * M * K + 3.0 * M * F_))      # states in, state gradients out, target in, pred + dpred out (dW: F x K)

    def timesum_bytes(inp, B_, T_, C, ld, out):
        return ("hbm timesum_kernel", 4.0 * B_ * C * (T_ + 1))             # C of ld columns of every (b, t) row in, (B, C) out

    saved = {n: kt.wrap(ops, n, fn) for n, fn in (("gru_seq_fwd", gru_flops("fwd")), ("gru_seq_bwd", gru_flops("bwd")),
                                                  ("gru_wide_fwd", gru_flops("fwd")), ("gru_wide_bwd", gru_flops("bwd")),
                                                  ("gru_coop_fwd", coop_flops("fwd")), ("gru_coop_bwd", coop_flops("bwd")),
                                                  ("gemm", gemm_flops), ("gemm_group", group_flops),
                                                  ("window_gather", gather_bytes), ("mse_fwd_bwd", mse_bytes), ("timesum", timesum_bytes),
                                                  ("head_stream", head_bytes))}
    # per-kernel durations are taken with the step's side-stream overlaps OFF (engine.set_overlap): a kernel that shares the chip with
    # another one is slower for reasons that are not its own.  The timed region above runs with them on.
    prev = model._engine.set_overlap(False)
    try:
        for _ in range(steps):
            win = loader.gather(loader.draw_starts())
            model.loss_step(win, 1.0, beta=1.0, kloss=Z, klmbda=0.1, bsize=B)
        agg = kt.summary()
    finally:
        model._engine.set_overlap(prev)
        for n, fn in saved.items():
            setattr(ops, n, fn)
    for d in agg.values():
        d["launches"] /= steps
        d["ms"] /= steps
        d["work"] /= steps
    return agg


NOMINAL_MHZ = 2400.0                      # MI355X_MICROARCH.md: the clock the 157.3 TF peak is quoted at


def at_clock(frac, mhz):
    """A fraction of the nominal peak restated against what the box's sustained shader clock allows: frac / (clock / 2400 MHz)."""
    return round(frac * NOMINAL_MHZ / mhz, 4) if mhz else None


def roofline_block(agg, value_per_gpu, mflop_per_window, ms_per_step, dump=False, region_mhz=None):
    mf = {k: d for k, d in agg.items() if not k.startswith("hbm ")}
    hb = {k: d for k, d in agg.items() if k.startswith("hbm ")}
    mfma_ms = sum(d["ms"] for d in mf.values())
    if dump:
        for k, d in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
            unit = "GB/s" if k.startswith("hbm ") else "TF"
            rate = d["work"] / max(d["ms"], 1e-9) / (1e6 if k.startswith("hbm ") else 1e9)
            print(f"{k:60s} x{d['launches']:4.0f} {d['ms']*1e3:9.1f} us/step {rate:8.1f} {unit}", file=sys.stderr)
    dom_key = max(mf, key=lambda k: mf[k]["ms"])
    dom = mf[dom_key]
    per_launch_ms = dom["ms"] / dom["launches"]
    achieved = dom["work"] / dom["launches"] / (per_launch_ms * 1e-3) / 1e12
    classes = {}
    for k, d in mf.items():
        c = k.split(" ")[0] + (" " + k.split(" ")[1] if k.startswith("gemm") else "")
        e = classes.setdefault(c, dict(ms=0.0, work=0.0, clk_ms=0.0, clk_w=0.0))
        e["ms"] += d["ms"]
        e["work"] += d["work"]
        e["clk_ms"] += d.get("clk_ms", 0.0)
        e["clk_w"] += d.get("clk_w", 0.0)
    by_class = {}
    for c, e in classes.items():
        tf = e["work"] / (e["ms"] * 1e-3) / 1e12
        mhz = e["clk_w"] / e["clk_ms"] if e["clk_ms"] > 0 else None
        by_class[c] = dict(ms_per_step=round(e["ms"], 3), tflops=round(tf, 2), clock_mhz=round(mhz, 0) if mhz else None,
                           frac_at_clock=at_clock(tf / PEAK_F32_MFMA_TFLOPS, mhz))
    for k, d in hb.items():                       # the bandwidth-bound kernels of the step against the 8 TB/s HBM roofline
        gbs = d["work"] / (d["ms"] * 1e-3) / 1e9
        by_class[k[4:]] = dict(bound="hbm", ms_per_step=round(d["ms"], 4), launches_per_step=d["launches"],
                               algorithmic_mb_per_step=round(d["work"] / 1e6, 2), gbs=round(gbs, 1), frac=round(gbs / PEAK_HBM_GBS, 4))
    step_frac = value_per_gpu * mflop_per_window * 1e6 / (PEAK_F32_MFMA_TFLOPS * 1e12)
    return dict(bound="mfma", kernel=dom_key, achieved=round(achieved, 3), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                frac=round(achieved / PEAK_F32_MFMA_TFLOPS, 4), traffic=pmc_traffic(dom_key),
                # the shader clock the box sustained WHILE the dominant kernel ran (vame_clock_stamp around each of its launches) and the
                # fraction of what that clock allows: frac mixes kernel quality with the box's power state, frac_at_clock does not
                clock_mhz=round(dom["mhz"], 0) if dom.get("mhz") else None, frac_at_clock=at_clock(achieved / PEAK_F32_MFMA_TFLOPS, dom.get("mhz")),
                launch_ms=round(per_launch_ms, 4), launches_per_step=dom["launches"],
                step_frac=round(step_frac, 4),
                clock_mhz_timed_region=round(region_mhz, 0) if region_mhz else None, step_frac_at_clock=at_clock(step_frac, region_mhz),
                timer_outliers=sum(d.get("outliers", 0) for d in agg.values()),
                by_class=by_class, timed_kernel_ms_per_step=round(sum(d["ms"] for d in agg.values()), 3),
                mfma_kernel_ms_per_step=round(mfma_ms, 3), non_mfma_ms_per_step=round(max(0.0, ms_per_step - mfma_ms), 3),
                # serial sum of every TIMED kernel minus the step time of the timed region: what the side-stream overlaps hide (positive).  Negative: the
                # launches this table does not time (split-K reductions, column sums, nuclear-norm solve, latent / loss / Adam kernels) plus launch gaps
                # outweigh what is hidden -- a host-bound small batch, or a wide shape (H > 256), where the engine runs without side streams at all
                hidden_by_overlaps_ms_per_step=round(sum(d["ms"] for d in agg.values()) - ms_per_step, 3),
                note="per-kernel times: 3 separate steps with the side-stream overlaps off (serial); ms_per_step: timed region with them on; "
                     "non_mfma_ms_per_step = max(0, ms_per_step - mfma_kernel_ms_per_step) understates the non-MFMA time by what the overlaps hide")


def lib_source_id():
    """sha256 of the kernel sources the loaded libvame_hip.so was built from (compiled in by the Makefile)."""
    from vame_amd import _lib
    return _lib.lib().vame_source_id().decode()


def pmc_traffic(dom_key):
    """HBM bytes per launch of a kernel from a committed rocprofv3 PMC summary (tools/rocprof_digest.py pmc: FETCH_SIZE x2 +
    WRITE_SIZE in separate --pmc passes, corrected as MI355X_MICROARCH.md prescribes) -- but only from a summary that was
    collected with a libvame_hip.so built from THESE kernel sources (vame_source_id(), stored in the summary); None otherwise: a
    number measured on an older kernel says nothing about the current one."""
    import glob
    sha = lib_source_id()
    # newest round first; a summary tools/run_profiles.sh has just collected in this very call (profiles/_this_run_*) before the committed ones
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_hbm_traffic*.json")), reverse=True)
    for path in sorted(paths, key=lambda p: not os.path.basename(p).startswith("_this_run_")):
        try:
            with open(path) as f:
                j = json.load(f)
        except (OSError, ValueError):
            continue
        if sha == "unidentified" or j.get("source_id") != sha:
            continue
        hit = j.get("by_bench_key", {}).get(dom_key)
        if hit is not None:
            return hit.get("hbm_bytes_per_launch_corrected")
    return None


# ------------------------------------------------------------------------------------------------ CPU baselines (oracle/torch_ref.py)
def _cpu_subprocess(fn_call, threads, timeout):
    """Run `oracle.torch_ref.<fn_call>` in a subprocess (own thread pool, hard time limit) and return its dict."""
    import subprocess
    code = ("import json,sys; sys.path.insert(0, %r); from oracle import torch_ref as R; "
            "print('CPUBASE ' + json.dumps(R.%s))" % (ROOT, fn_call.replace("THREADS", str(threads))))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout)
    for line in r.stdout.splitlines():
        if line.startswith("CPUBASE "):
            return json.loads(line[8:])
    raise RuntimeError(r.stderr[-500:])


def cpu_baseline(sample_steps=32):
    """Reference-equivalent torch-CPU train step (oracle/torch_ref.py) on this box's host cores, in a
    subprocess with a hard time limit (a bounded sample: 32 steps of B=256, about 10 s of CPU work, at the better of two
    thread counts; the other count is only probed with 4 steps)."""
    import subprocess
    avail = len(os.sched_getaffinity(0))
    best, tried = None, []
    for threads, steps in ((min(avail, 16), sample_steps), (min(avail, 64), min(4, sample_steps))):   # torch-CPU GRUs stop scaling early: report the better count
        if any(t == threads for t, _ in tried):
            continue
        try:
            d = _cpu_subprocess(f"time_train_steps(B=256, steps={steps}, warmup=2, threads=THREADS)", threads, 120)
            d["value"] = round(d["value"], 1)
            tried.append((threads, d["value"]))
            if best is None or d["value"] > best["value"]:
                best = d
        except subprocess.TimeoutExpired:
            print(f"cpu baseline with {threads} threads exceeded 120 s", file=sys.stderr)
    if best is not None:
        best["host_cpus_visible"] = avail
        best["thread_counts_tried"] = tried
        best["note"] = ("cores = the thread count at which stock torch-CPU runs this model fastest on this host (its nn.GRU stops "
                        "scaling around 16 threads; the larger count is probed and listed), not a choice to use few cores")
        return best
    return dict(value=None, unit="windows/s", cores=0, kind="port", sample="timed out on this host")


def cpu_baseline_embed():
    """CPU baseline of the embedding leg: the reference's batch-1 loop AS WRITTEN (pose_segmentation.py:87-98:
    one window per forward) and a batch-256 variant of the same model, both on a bounded sample (about 10 s each)."""
    import subprocess
    avail = len(os.sched_getaffinity(0))
    threads = min(avail, 16)
    out = {}
    for key, call in (("as_written_batch1", "time_embed(batch=1, budget_s=10.0, threads=THREADS)"),
                      ("batch256", "time_embed(batch=256, budget_s=10.0, threads=THREADS)")):
        try:
            d = _cpu_subprocess(call, threads, 120)
            d["value"] = round(d["value"], 1)
            out[key] = d
        except (subprocess.TimeoutExpired, RuntimeError) as e:
            out[key] = dict(value=None, sample=f"failed: {type(e).__name__}")
    best = max((d for d in out.values() if d.get("value")), key=lambda d: d["value"], default=None)
    return dict(value=best["value"] if best else None, unit="windows/s", cores=threads, kind="port",
                sample="embedding loop of oracle/torch_ref.py (stock torch nn.GRU encoder + mean head, eval mode): "
                       + "; ".join(f"{k}: {d.get('value')} windows/s ({d.get('sample')})" for k, d in out.items()),
                as_written_batch1=out["as_written_batch1"].get("value"), batch256=out["batch256"].get("value"),
                host_cpus_visible=avail)


# ------------------------------------------------------------------------------------------------ legs
class _SynthDataset:
    """Synthetic stand-in for SEQUENCE_DATASET: an already z-scored (F, N) series."""
    data_points = N_SERIES
    X = np.empty((F, 1))

    def __init__(self, T):
        self.temporal_window = 2 * T

    @staticmethod
    def normalised_f32():
        return synth_series(N_SERIES)


def timed_region(step, steps, world, sync, dev, info=None):
    """EXACTLY `steps` steps bracketed by barrier + device sync on both sides; the MAX over ranks.  `info` (dict) receives the
    average shader clock over the region on this rank (`mhz`, GPU only: vame_clock_stamp right behind the opening sync and in
    front of the closing one) and, with several ranks, every rank's own time before the closing barrier (`rank_dts`)."""
    from vame_amd import ops
    probe = ops.ClockProbe(dev) if dev.type == "cuda" else None
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    if probe is not None:
        probe.start()
    for _ in range(steps):
        terms = step()
    if probe is not None:
        probe.stop()
    sync()
    own = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        each = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(each, torch.tensor([own], device=dev, dtype=torch.float64))
        if info is not None:
            info["rank_dts"] = [float(e.item()) for e in each]
    if info is not None:
        info["mhz"] = probe.mhz() if probe is not None else None
    return dt, terms


def release_leg(on_gpu):
    """Between two legs of one bench process: drop everything the finished leg allocated (model, engine workspaces, the ops module's
    scratch caches) and hand the device memory back, so that every leg starts from the allocator state of a fresh process."""
    import gc
    from vame_amd import ops
    ops._ws_cache.clear()              # the module-level split-K / column-sum scratch grows to the largest leg's size and would stay
    ops._colsum_ws.clear()             # allocated in the middle of the next leg's address range (measured: the latency-bound batch-256
    gc.collect()                       # leg runs 2.70 ms/step in a fresh process, 2.75-2.87 behind a large leg, 2.70 again with these dropped)
    if on_gpu:
        torch.cuda.empty_cache()


def train_leg(dev, H, T, B, steps, warmup, rank, world, repeats=1, profile=True, dump=False, one_rank_leg=False, engine_options=None, graph=False):
    """Build the model at (H, T), run `warmup` untimed + `repeats` timed regions of `steps` steps.  Returns a dict with the region
    times, the last loss terms, the roofline block (GPU only) and -- several ranks -- the collective's own numbers."""
    from vame_amd.model.dataloader import DeviceWindowLoader
    from vame_amd.model.rnn_model import RNN_VAE
    from vame_amd.model.rnn_vae import FusedAdamAMSGrad, allreduce_gradients
    on_gpu = dev.type == "cuda"
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    torch.manual_seed(19)
    model = RNN_VAE(2 * T, Z, F, 1, FS, H, H, H, H, 0, 0, 0, False).to(dev).train()
    model.engine_options = dict(engine_options or {})
    opt = FusedAdamAMSGrad(model, lr=5e-4)
    loader = DeviceWindowLoader(_SynthDataset(T), B, T + FS, dev, rank=0, world=1)
    np.random.seed(1000 + rank)

    def step(reduce=True):
        win = loader.gather(loader.draw_starts())
        terms = model.loss_step(win, 1.0, beta=1.0, kloss=Z, klmbda=0.1, bsize=B)
        gs = allreduce_gradients(model) if reduce else 1.0
        opt.step(gscale=gs)
        return terms

    if graph and on_gpu and world == 1:       # the same step as ONE replayed hipGraph (what train() does up to batch 1024: rnn_vae.GraphedTrainStep)
        from vame_amd.model.rnn_vae import GraphedTrainStep
        gstep = GraphedTrainStep(model, opt, loader, torch.zeros(6, device=dev, dtype=torch.float64), kl_weight=1.0, beta=1.0, kloss=Z, klmbda=0.1, bsize=B)
        eager_step = step

        def step(reduce=True):                # noqa: F811
            return gstep(loader.draw_starts())

    for _ in range(warmup):
        terms = step()
    dts, infos = [], []
    for _ in range(repeats):
        info = {}
        dt, terms = timed_region(step, steps, world, sync, dev, info)
        dts.append(dt)
        infos.append(info)
    last = [float(v) for v in terms.cpu()]
    assert all(np.isfinite(last)), f"non-finite loss terms {last}"
    res = dict(dts=dts, last=last, mflop=train_mflop_per_window(H, T), roofline=None, distributed=None, mhz=infos[0].get("mhz"),
               rank_dts=infos[0].get("rank_dts"))
    if world > 1:
        res["distributed"] = collective_block(model, step, steps, rank, world, sync, dev, dts[0], one_rank_leg)
    if profile and on_gpu and rank == 0:
        agg = profile_kernels(model, loader, B)
        res["agg"] = agg
        res["roofline"] = roofline_block(agg, B * steps / dts[0], res["mflop"], dts[0] / steps * 1e3, dump, region_mhz=res["mhz"])
    del model, opt, loader
    release_leg(on_gpu)
    return res


def collective_block(model, step, steps, rank, world, sync, dev, dt_n, one_rank_leg):
    """Evidence that the N-rank line really went through the collective library: backend, library version, the all-reduce of the
    real gradient bucket timed on its own, and a 1-rank leg of the same step in the same run (rank 0 alone, no all-reduce) for
    the weak-scaling efficiency.  Every rank takes part in the collectives; rank 0 keeps the numbers."""
    bucket = model._flat_g_comm
    on_gpu = dev.type == "cuda"
    for _ in range(3):
        dist.all_reduce(bucket, op=dist.ReduceOp.SUM)
    dist.barrier()
    sync()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        dist.all_reduce(bucket, op=dist.ReduceOp.SUM)
    sync()
    ar_us = (time.perf_counter() - t0) / reps * 1e6
    tt = torch.tensor([ar_us], device=dev, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    ar_us = float(tt.item())
    nbytes = bucket.numel() * 4
    out = dict(world_size=dist.get_world_size(), backend=dist.get_backend(), allreduce_bucket_bytes=nbytes,
               allreduce_us=round(ar_us, 1), allreduce_algbw_gbs=round(nbytes / ar_us / 1e3, 2),
               allreduce_busbw_gbs=round(nbytes / ar_us / 1e3 * 2 * (world - 1) / world, 2),
               collective_library=(("RCCL (torch nccl backend) " + ".".join(str(v) for v in torch.cuda.nccl.version())) if on_gpu else "gloo (CPU test harness)"),
               devices=[torch.cuda.get_device_name(i) for i in range(torch.cuda.device_count())] if on_gpu else [])
    if one_rank_leg:
        dist.barrier()
        dt1 = None
        if rank == 0:                                  # the other ranks wait in the barrier below with an idle GPU
            step(reduce=False)
            sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                step(reduce=False)
            sync()
            dt1 = time.perf_counter() - t0
        dist.barrier()
        if rank == 0:
            out["one_rank_leg_ms_per_step"] = round(dt1 / steps * 1e3, 3)
            out["weak_scaling_eff"] = round(dt1 / dt_n, 4)          # same per-GPU work: t(1 rank) / t(N ranks)
    return out


def embed_leg(dev, n_win_per_rank, rank, world, H=256, T=30, engine_options=None):
    """Encoder-only latent embedding of a synthetic series, window index range sharded over ranks (no collective on the data path)."""
    from vame_amd.analysis.pose_segmentation import embed_series
    from vame_amd.model.rnn_model import RNN_VAE
    torch.manual_seed(19)
    model = RNN_VAE(2 * T, Z, F, 1, FS, H, H, H, H, 0, 0, 0, False).to(dev).eval()
    model.engine_options = dict(engine_options or {})
    n_win = n_win_per_rank * world
    data = synth_series(n_win + T)
    embed_series(model, data[:, :70000], batch=16384)                      # warm-up (allocations, clocks)
    if world > 1:
        dist.barrier()
    if dev.type == "cuda":
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    probe = None
    if dev.type == "cuda":
        from vame_amd import ops
        probe = ops.ClockProbe(dev)
        probe.start()
    out, (lo, hi) = embed_series(model, data, batch=16384, rank=rank, world=world)
    if dev.type == "cuda":
        probe.stop()
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(out).all()
    value = n_win / dt
    # encoder (two bi-GRU layers) + both Lambda heads, matmul flops, 2 / MAC: 96.707 MFLOP per window at H = 256, T = 30
    mflop = (2 * T * 2 * 3 * H * (F + H) + 2 * T * 2 * 3 * H * (2 * H + H) + 2 * 2 * 4 * H * Z) / 1e6
    assert (H, T) != (256, 30) or abs(mflop - MFLOP_PER_WINDOW_EMBED) < 1e-3
    tf = value / world * mflop * 1e6 / 1e12
    mhz = probe.mhz() if probe is not None else None
    del model, out
    release_leg(dev.type == "cuda")
    return dict(metric=f"latent-embedding windows/sec (encoder + Lambda mean) T={T},F={F},h={H}", value=round(value, 1), unit="windows/s",
                n_gpus=world, seconds=round(dt, 3), windows=n_win,
                includes="host->device upload of the series + window gather + encoder + mean",
                config=dict(workload=f"BASELINE.json configs[4] shape: {n_win_per_rank} stride-1 windows per GPU, batch 16384",
                            parallelism=f"shard{world}"),
                roofline=dict(bound="mfma", unit="TFLOP/s", peak=PEAK_F32_MFMA_TFLOPS, achieved=round(tf, 2),
                              frac=round(tf / PEAK_F32_MFMA_TFLOPS, 4), clock_mhz=round(mhz, 0) if mhz else None,
                              frac_at_clock=at_clock(tf / PEAK_F32_MFMA_TFLOPS, mhz),
                              traffic=pmc_traffic("gru_seq_fwd_kernel<256> x2 streams gi T=30 embed") if dev.type == "cuda" else None))


PEAK_BF16_MFMA_TFLOPS = 2500.0             # MI355X_MICROARCH.md: dense bf16 MFMA peak


def split_gemm_block(dev, steps, warmup, default_res):
    """OPT-IN leg (not the headline): the same configs[1] train step with the large weight gradients (the six dW_hh, the two layer-1 dW_ih,
    the future decoder's two dW_hh: two k-major operands, K = batch x time; vame_gemm_group_bf16x6_f32, engine option split_wgrad) and the
    second encoder layer's input projections and data gradients (row-major activations x weights, K = a layer width; vame_gemm_bf16x6_f32,
    engine option split_proj) on the error-compensated split-bf16 contraction instead of the f32-input matrix cores; everything else
    unchanged.  fp32-equivalent flops against the bf16 dense peak / 6 plane products."""
    out = dict(arith="large weight gradients, layer-1 input projections and their data gradients: bf16x6 split operands (three exact bf16 planes per fp32 "
                     "value, six plane products), f32 accumulate; everything else (the GRU recurrences, BPTT, the small contractions): f32-input MFMA as in the headline",
               peak=round(PEAK_BF16_MFMA_TFLOPS / 6, 1), unit="TFLOP/s (fp32-equivalent)", peak_note="bf16 dense peak 2500 TF / 6 plane products",
               default_ms_per_step=round(default_res["dts"][0] / steps * 1e3, 3),
               error_table="profiles/r05_split_gemm_error_table.txt (max error vs float64: 0.4x the f32-input kernel's with two accumulators per output, "
                           "1.0x with one); parity: tests/test_model_gpu.py::test_headline_batch_4096_all_gradients... at the unchanged tolerance, both ways")
    dom = "gemm_kernel TN M=768 N=256 K=122880 x6 grouped"
    base = (default_res.get("agg") or {}).get(dom)
    if base:
        out["default_launch"] = dict(kernel=dom, launch_ms=round(base["ms"] / base["launches"], 4),
                                     tflops=round(base["work"] / base["launches"] / (base["ms"] / base["launches"] * 1e-3) / 1e12, 1))
    for name, optw in (("one_accumulator", 1), ("two_accumulators", 0)):
        r = train_leg(dev, 256, 30, 4096, steps, warmup, 0, 1, engine_options=dict(split_wgrad=optw, split_proj=optw))
        line = dict(engine_option=f"split_wgrad={optw} split_proj={optw}", value=round(4096 * steps / r["dts"][0], 1), unit="windows/s",
                    ms_per_step=round(r["dts"][0] / steps * 1e3, 3), clock_mhz_timed_region=round(r["mhz"], 0) if r["mhz"] else None)
        launches = {}
        for k, d in (r.get("agg") or {}).items():
            if " bf16x6 " in k:
                tf = d["work"] / d["launches"] / (d["ms"] / d["launches"] * 1e-3) / 1e12
                launches[k] = dict(launch_ms=round(d["ms"] / d["launches"], 4), tflops=round(tf, 1), frac=round(tf / (PEAK_BF16_MFMA_TFLOPS / 6), 4),
                                   clock_mhz=round(d["mhz"], 0) if d.get("mhz") else None)
        line["launches"] = launches
        out[name] = line
    # the stock batch (256) as a replayed hipGraph with both options (K = 7,680 for the weight gradients, M = 7,680 for the projections)
    r = train_leg(dev, 256, 30, 256, 200, 30, 0, 1, profile=False, engine_options=dict(split_wgrad=1, split_proj=1), graph=True)
    out["batch256_one_accumulator"] = dict(engine_option="split_wgrad=1 split_proj=1", execution="one replayed hipGraph per step", value=round(256 * 200 / r["dts"][0], 1),
                                           unit="windows/s", ms_per_step=round(r["dts"][0] / 200 * 1e3, 3))
    # the embedding leg with its one large contraction (the layer-1 input projection, two directions) on the split form
    e = embed_leg(dev, 2_000_000, 0, 1, engine_options=dict(split_proj=1))
    out["embed_one_accumulator"] = dict(engine_option="split_proj=1", value=e["value"], unit="windows/s", seconds=e["seconds"], windows=e["windows"],
                                        clock_mhz=e["roofline"]["clock_mhz"])
    return out


def workload_name(H, T, B, world):
    if world > 1:
        head = f"BASELINE.json configs[2] shape on {world} GPUs (data-parallel)"
    elif (H, T, B) == (256, 30, 4096):
        head = "BASELINE.json configs[1]"
    elif (H, T, B) == (512, 60, 8192):
        head = "BASELINE.json configs[3]"
    elif (H, T, B) == (256, 30, 256):
        head = "BASELINE.json configs[0] shape on the GPU (the reference's stock config.yaml batch, vame/initialize_project/new.py:111)"
    else:
        head = "non-headline shape (BASELINE.json configs[3] is H=512,T=60,batch 8192)"
    return head + f": T={T},F={F},zdims={Z},hidden={H},FS={FS}, batch={B}/GPU fp32 train step (gather+fwd+loss+bwd+allreduce+Adam-amsgrad)"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-steps", type=int, default=None,
                    help="train steps (B=256) of the CPU baseline sample; default 32 on a GPU box.  Naming it also runs the baseline "
                         "where nothing else is measured (the CPU test-suite's emulator harness)")
    ap.add_argument("--no-also", action="store_true", help="skip the configs[3] / configs[4] legs and the two extra timed regions")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--graph", action="store_true", help="run the step as one replayed hipGraph (single GPU; train() does this up to batch 1024)")
    ap.add_argument("--dump-kernels", action="store_true", help="per-launch-group table on stderr")
    ap.add_argument("--mode", choices=["train", "embed"], default="train",
                    help="train = the headline metric; embed = encoder-only embedd_latent_vectors sweep (BASELINE config 5)")
    ap.add_argument("--embed-windows", type=int, default=2_000_000)
    ap.add_argument("--hidden", type=int, default=256, help="exploration only (BASELINE config 4 uses 512)")
    ap.add_argument("--time-window", type=int, default=30, help="exploration only (BASELINE config 4 uses 60)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)                            # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with --nproc-per-node {args.gpus} "
                         f"(or run `python bench.py --gpus {args.gpus}` without a torchrun environment: it starts the ranks itself)")
    from vame_amd import _lib
    from vame_amd.model.rnn_vae import _maybe_init_distributed
    dev = _lib.device(local)                         # raises without an MI355X: the measured path has no CPU fallback
    on_gpu = dev.type == "cuda"                      # (False only under the CPU test-suite's harness, tests/emu/harness.py, which
    if on_gpu and torch.cuda.device_count() < world:                 # checks the rank / JSON plumbing and measures nothing)
        raise SystemExit(f"bench.py: --gpus {world} but only {torch.cuda.device_count()} GPU(s) are visible")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        _maybe_init_distributed()                    # one process per GPU: backend nccl (= RCCL over xGMI)
        assert dist.get_world_size() == world

    H, T, B = args.hidden, args.time_window, args.batch
    if args.mode == "embed":
        out = embed_leg(dev, args.embed_windows, rank, world, H, T)
        if rank == 0:
            out.update(higher_is_better=True, scaling="weak", dtype="f32", data="synthetic")
            if not args.no_cpu_baseline and world == 1 and on_gpu:
                out["cpu_baseline"] = cpu_baseline_embed()
            print(json.dumps(out))
    else:
        headline = (H, T, B) == (256, 30, 4096)
        extras = headline and not args.no_also
        res = train_leg(dev, H, T, B, args.steps, args.warmup, rank, world, repeats=3 if extras else 1,
                        dump=args.dump_kernels, one_rank_leg=True, graph=args.graph)
        also = None
        if extras and world == 1 and on_gpu:
            # the other two single-GPU configurations of BASELINE.json under the same clock, each with its own roofline, and the batch
            # the reference's stock config.yaml trains at (256: vame/initialize_project/new.py:111 -- the configuration real users run)
            c4 = train_leg(dev, 512, 60, 8192, 5, 2, 0, 1)
            c4_line = dict(metric="temporal windows/sec (train) T=60,F=24,h=512", value=round(8192 * 5 / c4["dts"][0], 1), unit="windows/s",
                           steps=5, warmup=2, ms_per_step=round(c4["dts"][0] / 5 * 1e3, 3), config=dict(workload=workload_name(512, 60, 8192, 1)),
                           roofline=c4["roofline"])
            b256 = train_leg(dev, 256, 30, 256, 30, 10, 0, 1, graph=True)
            b256e = train_leg(dev, 256, 30, 256, 30, 10, 0, 1, profile=False)
            # train() with `vame_amd_hip_graph: auto` times both forms in its first epoch and keeps the faster (GraphedTrainStep._measure): so does this leg
            g_ms, e_ms = b256["dts"][0] / 30 * 1e3, b256e["dts"][0] / 30 * 1e3
            pick = "graph" if g_ms <= e_ms else "eager"
            b256_line = dict(metric="temporal windows/sec (train) T=30,F=24,h=256", value=round(256 / min(g_ms, e_ms) * 1e3, 1), unit="windows/s",
                             steps=30, warmup=10, ms_per_step=round(min(g_ms, e_ms), 3),
                             execution=f"the faster of the two forms on this box, as train()'s hip_graph 'auto' picks it: {pick}",
                             graph=dict(value=round(256 / g_ms * 1e3, 1), ms_per_step=round(g_ms, 3),
                                        execution="one hipGraph replay per step (rnn_vae.GraphedTrainStep)"),
                             eager=dict(value=round(256 / e_ms * 1e3, 1), ms_per_step=round(e_ms, 3), execution="the same step enqueued launch by launch"),
                             config=dict(workload=workload_name(256, 30, 256, 1)), roofline=b256["roofline"])
            emb_line = embed_leg(dev, 2_000_000, 0, 1)
            if not args.no_cpu_baseline:              # the reference's batch-1 loop as written and a batch-256 variant, beside the embedding number
                emb_line["cpu_baseline"] = cpu_baseline_embed()
            also = dict(configs3_h512_t60_b8192=c4_line, configs4_embed_1gpu=emb_line, batch256=b256_line,
                        split_gemm=split_gemm_block(dev, args.steps, args.warmup, res))
        out = None
        if rank == 0:
            dt = res["dts"][0]
            value = B * world * args.steps / dt
            out = dict(metric=f"temporal windows/sec (train) T={T},F={F},h={H}", value=round(value, 1), unit="windows/s", n_gpus=world,
                       steps=args.steps, warmup=args.warmup, ms_per_step=round(dt / args.steps * 1e3, 3), higher_is_better=True,
                       scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
                       config=dict(workload=workload_name(H, T, B, world), global_batch=B * world, parallelism=f"dp{world}",
                                   last_loss_terms=res["last"]),
                       roofline=res["roofline"])
            if res["rank_dts"] is not None:          # every rank's own time for the same region (before the closing barrier): `value` uses the MAX
                per = [d / args.steps * 1e3 for d in res["rank_dts"]]
                out["ms_per_step_by_rank"] = dict(min=round(min(per), 3), max=round(max(per), 3), ranks=[round(v, 3) for v in per])
            if len(res["dts"]) > 1:                  # `value` is the first region; the other two only show the spread
                vals = sorted(B * world * args.steps / d for d in res["dts"])
                out["repeat_spread"] = dict(regions=len(vals), steps_each=args.steps, min=round(vals[0], 1), median=round(vals[len(vals) // 2], 1),
                                            max=round(vals[-1], 1), unit="windows/s")
            if res["distributed"] is not None:
                out["distributed"] = res["distributed"]
            if also is not None:
                out["also"] = also
        if world > 1:                                # the ranks part here: the CPU baseline below is rank 0's alone, nobody waits in a collective for it
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            if not args.no_cpu_baseline and (on_gpu or args.cpu_baseline_steps is not None):
                out["cpu_baseline"] = cpu_baseline(args.cpu_baseline_steps or 32)     # rank 0's host cores, N = 1 and N > 1 alike
            print(json.dumps(out))
        return
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
