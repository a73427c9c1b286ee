#!/usr/bin/env python3
"""Benchmark of the VAME RNN-VAE training hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic temporal windows: window gather
(sliding-window batcher) -> RNN-VAE forward -> MSE + future-MSE + KL + nuclear-norm loss -> BPTT
-> one RCCL all-reduce of the flat gradient bucket (N > 1) -> fused Adam-AMSGrad.  Workload =
BASELINE.json config 2: T=30, F=24, H=256, Z=30, FS=15, batch 4096 windows per GPU, fp32.
The series is already resident in HBM when the timed region starts.

Prints ONE JSON line (rank 0): metric/value/unit + roofline (dominant kernel, measured with HIP
events on the launch stream) + cpu_baseline (reference-equivalent torch-CPU model, oracle/torch_ref.py).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

T, F, Z, H, FS = 30, 24, 30, 256, 15
B_LOCAL = 4096
N_SERIES = 1_000_000
MFLOP_PER_WINDOW_TRAIN = 400.343          # matmul flops, 2/MAC, bwd = 2x fwd, decoder input projection once
PEAK_F32_MFMA_TFLOPS = 157.3              # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0


def synth_series(n, seed=0):
    rng = np.random.default_rng(seed)
    i = np.arange(n, dtype=np.float64)
    X = np.sin(2 * np.pi * i[None, :] * (np.arange(F, dtype=np.float64)[:, None] + 1) / 997.0) + 0.5 * rng.standard_normal((F, n))
    return ((X - X.mean()) / X.std()).astype(np.float32)


class KernelTimer:
    """HIP-event timing of each C-ABI launch group on the stream it is launched on (torch's current stream)."""

    def __init__(self):
        self.records = []

    def wrap(self, ops_mod, name, flops_fn):
        inner = getattr(ops_mod, name)

        def timed(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = inner(*a, **k)
            e1.record()
            self.records.append((name, flops_fn(*a, **k), e0, e1))
            return r
        setattr(ops_mod, name, timed)
        return inner

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for name, (key, flops), e0, e1 in self.records:
            d = agg.setdefault(key, dict(api=name, launches=0, ms=0.0, flops=0.0))
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += flops
        return agg


def profile_kernels(model, loader, steps=3):
    """Per-kernel-class time + algorithmic flops over `steps` train steps (separate from the timed region)."""
    from vame_amd import ops
    kt = KernelTimer()

    def gru_flops(tag):
        def f(streams, B, Hh):
            key_t = ops.GF["T"] if tag == "fwd" else ops.GB["T"]
            fl = sum(2.0 * 3 * Hh * Hh * B * int(s[key_t]) for s in streams)
            name = "gru_seq" if Hh <= 256 else "gru_wide"
            return (f"{name}_{tag}_kernel<{Hh}> x{len(streams)} streams", fl)
        return f

    def gemm_flops(M, N, K, A, akm, Bm, bkm, *a, **k):
        kind = "NT" if (not akm and not bkm) else ("NN" if not akm else "TN")
        return (f"gemm_kernel {kind} M={M} N={N} K={K}", 2.0 * M * N * K)
    def group_flops(M, N, K, As, akm, Bs, bkm, *a, **k):
        kind = "NT" if (not akm and not bkm) else ("NN" if not akm else "TN")
        return (f"gemm_kernel {kind} M={M} N={N} K={K} x{len(As)} grouped", 2.0 * M * N * K * len(As))
    saved = {n: kt.wrap(ops, n, fn) for n, fn in (("gru_seq_fwd", gru_flops("fwd")), ("gru_seq_bwd", gru_flops("bwd")),
                                                  ("gru_wide_fwd", gru_flops("fwd")), ("gru_wide_bwd", gru_flops("bwd")),
                                                  ("gemm", gemm_flops), ("gemm_group", group_flops))}
    try:
        for _ in range(steps):
            win = loader.gather(loader.draw_starts())
            model.loss_step(win, 1.0, beta=1.0, kloss=Z, klmbda=0.1, bsize=B_LOCAL)
        agg = kt.summary()
    finally:
        for n, fn in saved.items():
            setattr(ops, n, fn)
    for d in agg.values():
        d["launches"] /= steps
        d["ms"] /= steps
        d["flops"] /= steps
    return agg


def bench_embed(args, dev, rank, world):
    """Encoder-only latent embedding of a synthetic series, window index range sharded over ranks (no collective)."""
    from vame_amd.analysis.pose_segmentation import embed_series
    from vame_amd.model.rnn_model import RNN_VAE
    model = RNN_VAE(2 * T, Z, F, 1, FS, H, H, H, H, 0, 0, 0, False).to(dev).eval()
    n_win = args.embed_windows * world
    data = synth_series(n_win + T)
    embed_series(model, data[:, :70000], batch=16384)                      # warm-up (allocations, clocks)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out, (lo, hi) = embed_series(model, data, batch=16384, rank=rank, world=world)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(out).all()
    if rank == 0:
        value = n_win / dt
        out = dict(metric="latent-embedding windows/sec (encoder + Lambda mean) T=30,F=24,h=256", value=round(value, 1),
                   unit="windows/s", n_gpus=world, higher_is_better=True, scaling="weak", dtype="f32", data="synthetic",
                   seconds=round(dt, 3), includes="host->device upload of the series + window gather + encoder + mean",
                   config=dict(workload=f"BASELINE.json configs[4] shape: {args.embed_windows} stride-1 windows per GPU, batch 16384",
                               parallelism=f"shard{world}"),
                   roofline=dict(bound="mfma", unit="TFLOP/s", peak=PEAK_F32_MFMA_TFLOPS,
                                 achieved=round(value / world * 96.707e6 / 1e12, 2),
                                 frac=round(value / world * 96.707e6 / 1e12 / PEAK_F32_MFMA_TFLOPS, 4), traffic=None))
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline_embed()
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def lib_source_id():
    """sha256 of the kernel sources the loaded libvame_hip.so was built from (compiled in by the Makefile)."""
    from vame_amd import _lib
    return _lib.lib().vame_source_id().decode()


def pmc_traffic(dom_key):
    """HBM bytes per launch of the dominant kernel from a committed rocprofv3 PMC summary (tools/pmc_summary.py: FETCH_SIZE x2 +
    WRITE_SIZE in separate --pmc passes, corrected as MI355X_MICROARCH.md prescribes) -- but only from a summary that was
    collected with a libvame_hip.so built from THESE kernel sources (vame_source_id(), stored in the summary); None otherwise: a number measured on an
    older kernel says nothing about the current one."""
    import glob
    sha = lib_source_id()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_hbm_traffic*.json")), reverse=True):
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


def cpu_baseline():
    """Reference-equivalent torch-CPU train step (oracle/torch_ref.py) on this box's host cores, in a
    subprocess with a hard time limit (a bounded sample: 32 steps of B=256, about 10 s of CPU work, at the better of two
    thread counts; the other count is only probed with 4 steps)."""
    import subprocess
    avail = len(os.sched_getaffinity(0))
    best, tried = None, []
    for threads, steps in ((min(avail, 16), 32), (min(avail, 64), 4)):   # torch-CPU GRUs stop scaling early: report the better count
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


def main():
    global B_LOCAL, H, T, MFLOP_PER_WINDOW_TRAIN
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=B_LOCAL)
    ap.add_argument("--dump-kernels", action="store_true", help="per-launch-group table on stderr")
    ap.add_argument("--mode", choices=["train", "embed"], default="train",
                    help="train = the headline metric; embed = encoder-only embedd_latent_vectors sweep (BASELINE config 5)")
    ap.add_argument("--embed-windows", type=int, default=2_000_000)
    ap.add_argument("--hidden", type=int, default=H, help="exploration only (BASELINE config 4 uses 512)")
    ap.add_argument("--time-window", type=int, default=T, help="exploration only (BASELINE config 4 uses 60)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    from vame_amd import _lib
    from vame_amd.model.rnn_vae import _maybe_init_distributed
    dev = _lib.device(local)                         # raises without an MI355X: the measured path has no CPU fallback
    on_gpu = dev.type == "cuda"                      # (False only under the CPU test-suite's harness, tests/emu/harness.py, which
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)      # checks the rank / JSON plumbing and measures nothing)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        _maybe_init_distributed()                    # one process per GPU: backend nccl (= RCCL over xGMI)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from vame_amd.model.dataloader import DeviceWindowLoader
    from vame_amd.model.rnn_model import RNN_VAE
    from vame_amd.model.rnn_vae import FusedAdamAMSGrad, allreduce_gradients

    B_LOCAL = args.batch
    if (args.hidden, args.time_window) != (H, T):          # non-headline shape: recompute the algorithmic flops per window
        H, T = args.hidden, args.time_window
        fwd = 2 * T * 2 * 3 * H * (F + H) + 2 * T * 2 * 3 * H * (2 * H + H) + 2 * 2 * 4 * H * Z     # encoder L0 + L1, Lambda
        for st in (T, FS):                                                                         # decoder, future decoder
            fwd += 2 * Z * 2 * H + 2 * 2 * 3 * H * Z + 2 * st * 2 * 3 * H * H + 2 * st * 2 * H * F
        MFLOP_PER_WINDOW_TRAIN = 3 * fwd / 1e6
    torch.manual_seed(19)
    if args.mode == "embed":
        return bench_embed(args, dev, rank, world)
    model = RNN_VAE(2 * T, Z, F, 1, FS, H, H, H, H, 0, 0, 0, False).to(dev).train()
    opt = FusedAdamAMSGrad(model, lr=5e-4)

    class _DS:   # synthetic stand-in for SEQUENCE_DATASET: already z-scored (F,N) series
        data_points, temporal_window = N_SERIES, 2 * T
        X = np.empty((F, 1))

        @staticmethod
        def normalised_f32():
            return synth_series(N_SERIES)
    loader = DeviceWindowLoader(_DS, B_LOCAL, T + FS, dev, rank=0, world=1)
    np.random.seed(1000 + rank)

    def step():
        win = loader.gather(loader.draw_starts())
        terms = model.loss_step(win, 1.0, beta=1.0, kloss=Z, klmbda=0.1, bsize=B_LOCAL)
        gs = allreduce_gradients(model)
        opt.step(gscale=gs)
        return terms

    for _ in range(args.warmup):
        terms = step()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        terms = step()
    sync()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    last = [float(v) for v in terms.cpu()]
    assert all(np.isfinite(last)), f"non-finite loss terms {last}"

    if rank == 0:
        value = B_LOCAL * world * args.steps / dt
        roof = None
        if on_gpu:
            agg = profile_kernels(model, loader)
            total_ms = sum(d["ms"] for d in agg.values())
            if args.dump_kernels:
                for k, d in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
                    print(f"{k:60s} x{d['launches']:4.0f} {d['ms']*1e3:9.1f} us/step {d['flops']/max(d['ms'],1e-9)/1e9:7.1f} TF", file=sys.stderr)
            dom_key = max(agg, key=lambda k: agg[k]["ms"])
            dom = agg[dom_key]
            per_launch_ms = dom["ms"] / dom["launches"]
            achieved = dom["flops"] / dom["launches"] / (per_launch_ms * 1e-3) / 1e12
            classes = {}
            for k, d in agg.items():
                c = k.split(" ")[0] + (" " + k.split(" ")[1] if k.startswith("gemm") else "")
                e = classes.setdefault(c, dict(ms=0.0, flops=0.0))
                e["ms"] += d["ms"]
                e["flops"] += d["flops"]
            roof = dict(bound="mfma", kernel=dom_key, achieved=round(achieved, 3), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                        frac=round(achieved / PEAK_F32_MFMA_TFLOPS, 4), traffic=pmc_traffic(dom_key),
                        launch_ms=round(per_launch_ms, 4), launches_per_step=dom["launches"],
                        step_frac=round(value / world * MFLOP_PER_WINDOW_TRAIN * 1e6 / (PEAK_F32_MFMA_TFLOPS * 1e12), 4),
                        by_class={c: dict(ms_per_step=round(e["ms"], 3), tflops=round(e["flops"] / (e["ms"] * 1e-3) / 1e12, 2))
                                  for c, e in classes.items()},
                        timed_kernel_ms_per_step=round(total_ms, 3))
        out = dict(metric=f"temporal windows/sec (train) T={T},F={F},h={H}", value=round(value, 1), unit="windows/s", n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=round(dt / args.steps * 1e3, 3), higher_is_better=True,
                   scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
                   config=dict(workload=((("BASELINE.json configs[1]" if (H, T, B_LOCAL) == (256, 30, 4096) else ("BASELINE.json configs[3]" if (H, T, B_LOCAL) == (512, 60, 8192) else "non-headline shape (BASELINE.json configs[3] is H=512,T=60,batch 8192)"))
                                          if world == 1 else f"BASELINE.json configs[2] shape on {world} GPUs (data-parallel)")
                                         + f": T={T},F={F},zdims={Z},hidden={H},FS={FS}, batch={B_LOCAL}/GPU fp32 train step "
                                         "(gather+fwd+loss+bwd+allreduce+Adam-amsgrad)"), global_batch=B_LOCAL * world,
                               parallelism=f"dp{world}", last_loss_terms=last),
                   roofline=roof)
        if not args.no_cpu_baseline and world == 1 and on_gpu:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
