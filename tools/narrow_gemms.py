"""Every contraction of one train step that has a dimension <= 32 (incl., since round 6, the streaming output heads and the grouped projections of z) (bench.py's "narrow" HBM rows), listed by shape: launches per step, time, algorithmic
bytes (A, B read once + C written once) and the rate against the 8 TB/s HBM roofline.  HIP-event times: launches shorter than the host's enqueue time
(~20 us with the event pair) read long here -- profiles/r05_kernel_stats.csv has their kernel-trace durations (the sum is ~0.78 ms per headline step
against ~1.03 ms by events); at ~4.5 TB/s for the six >= 130 MB launches the step would gain ~0.15 ms (1 %).

    python tools/narrow_gemms.py [B=4096] [H=256] [T=30]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                        # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    from vame_amd import ops
    from vame_amd.model.dataloader import DeviceWindowLoader
    from vame_amd.model.rnn_model import RNN_VAE
    dev = torch.device("cuda", 0)
    torch.manual_seed(19)
    model = RNN_VAE(2 * T, bench.Z, bench.F, 1, bench.FS, H, H, H, H, 0, 0, 0, False).to(dev).train()
    loader = DeviceWindowLoader(bench._SynthDataset(T), B, T + bench.FS, dev, rank=0, world=1)
    np.random.seed(1000)

    def step():
        model.loss_step(loader.gather(loader.draw_starts()), 1.0, beta=1.0, kloss=bench.Z, klmbda=0.1, bsize=B)

    for _ in range(3):
        step()
    kt = bench.KernelTimer()

    def kind(akm, bkm):
        return "NT" if (not akm and not bkm) else ("NN" if not akm else "TN")

    def one(M, N, K, A, akm, Bm, bkm, *a, **k):
        if min(M, N, K) > 32:
            return ("wide", 0.0)
        return (f"{kind(akm, bkm)} M={M} N={N} K={K} acc={int(bool(k.get('accumulate', a[4] if len(a) > 4 else False)))}", 4.0 * (M * K + K * N + M * N))

    def group(M, N, K, As, akm, Bs, bkm, *a, **k):
        if min(M, N, K) > 32:
            return ("wide", 0.0)
        return (f"{kind(akm, bkm)} M={M} N={N} K={K} x{len(As)} grouped", 4.0 * (M * K + K * N + M * N) * len(As))

    def head(Yop, M, F_, K, *a, **k):                               # round 6: the streaming output head (prediction + MSE + dY + dW in one pass over the states)
        return (f"head_stream M={M} F={F_} K={K} (+ its dW reduction)", 4.0 * (2.0 * M * K + 3.0 * M * F_))

    def lin(A, M, K, problems):                                     # round 6: the decoders' projections of z in one launch
        return (f"linear_group M={M} K={K} N={'+'.join(str(p[4]) for p in problems)}", 4.0 * (M * K + sum(p[4] * (K + M) for p in problems)))

    saved = {n: kt.wrap(ops, n, f) for n, f in (("gemm", one), ("gemm_group", group), ("head_stream", head), ("linear_group", lin))}
    prev = model._engine.set_overlap(False)
    steps = 5
    for _ in range(steps):
        step()
    agg = kt.summary()
    model._engine.set_overlap(prev)
    for n, f in saved.items():
        setattr(ops, n, f)
    tot = 0.0
    print(f"narrow contractions of one train step, batch {B}, H {H}, T {T} (overlaps off)")
    for key, d in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
        if key == "wide":
            continue
        ms, n, w = d["ms"] / steps, d["launches"] / steps, d["work"] / steps
        tot += ms
        print(f"{key:48s} x{n:4.1f} {ms * 1e3:8.1f} us/step {w / 1e6:9.1f} MB {w / ms / 1e6:8.0f} GB/s  frac {w / ms / 1e6 / 8000:.3f}")
    print(f"total {tot * 1e3:.1f} us/step")


if __name__ == "__main__":
    main()
