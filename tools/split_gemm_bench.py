#!/usr/bin/env python3
"""Error table and timing of the error-compensated split-bf16 contraction (vame_gemm_group_bf16x6_f32) beside the f32-input MFMA kernel
(vame_gemm_group_f32) on the weight-gradient shapes, and of its row-major-A form (vame_gemm_bf16x6_f32 beside vame_gemm_f32) on the layer-1
projection / data-gradient shapes, against a float64 product.

    python tools/split_gemm_bench.py [--reps 10] [--quick]

Columns: max over outputs of |C - C64| / sum_k |a||b|  (error in units of the products' magnitude: what a summation error analysis bounds)
and max |C - C64| / max |C64| (relative to the tensor's own scale: the unit of tests/tolerances.py), for the f32 kernel and for the split
kernel with two / one accumulators per output; then the launch time (HIP events around `reps` launches incl. the split-K reduction).
Data: N(0,1) x a log-uniform row scale 2^[-8, 8] (gradient-like dynamic range) unless noted."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vame_amd import ops  # noqa: E402
from vame_amd.ops import Operand  # noqa: E402


def run(M, N, K, n, sk, gap, seg, reps, wide=True, dev="cuda"):
    g = torch.Generator(device=dev).manual_seed(7)
    Mw = M + gap
    gap_at = (M // 2) // 4 * 4 if gap else 0

    def mk(width):
        v = torch.randn(K, width, device=dev, generator=g)
        if wide:
            v *= torch.exp2(torch.empty(K, 1, device=dev).uniform_(-8, 8, generator=g))
        return v
    A, Bm = [mk(Mw) for _ in range(n)], [mk(N) for _ in range(n)]
    if seg:                                  # B in the (batch, T + 2, 2H) sequence layout (two-level rows), like h_{t-1} of the engine
        Bst = [torch.zeros(K // seg, seg + 2, 2 * N, device=dev) for _ in range(n)]
        for s, b in zip(Bst, Bm):
            s[:, 1:seg + 1, :N] = b.view(K // seg, seg, N)
        opB = [Operand(s, 2 * N, off=2 * N, seg=seg, seg_stride=(seg + 2) * 2 * N) for s in Bst]
    else:
        opB = [Operand(b, N) for b in Bm]
    opA = [Operand(a, Mw) for a in A]
    cols = torch.cat([torch.arange(0, gap_at), torch.arange(gap_at + gap, Mw)]).to(dev) if gap else torch.arange(M, device=dev)
    C = torch.zeros(n, M, N, device=dev)
    ws = torch.empty(n * sk * M * N, device=dev)
    offs = [i * M * N for i in range(n)]
    a64 = A[0][:, cols].double()
    ref = a64.T @ Bm[0].double()
    mag = a64.abs().T @ Bm[0].double().abs()
    row = {}
    for name, split in (("f32", None), ("split2", 0), ("split1", 1)):
        if split is not None and not ops.gemm_split_ok(M, N, K, opA, opB, sk, gap_at, gap):
            row[name] = None
            continue
        call = lambda: ops.gemm_group(M, N, K, opA, 1, opB, 1, C, offs, N, sk, ws, a_gap_at=gap_at, a_gap=gap, split=split)  # noqa: E731
        C.zero_()
        call()
        torch.cuda.synchronize()
        err = (C[0].double() - ref).abs()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        call()
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        row[name] = (float((err / mag).max()), float(err.max() / ref.abs().max()), ms, 2.0 * M * N * K * n / ms / 1e9)
    return row


def run_rows(M, N, K, seg, bkm, reps, dev="cuda"):
    """vame_gemm_bf16x6_f32 (row-major A = activations in the engine's sequence layout, B = a weight matrix) beside vame_gemm_f32."""
    g = torch.Generator(device=dev).manual_seed(9)
    a = torch.randn(M, K, device=dev, generator=g) * torch.exp2(torch.empty(M, 1, device=dev).uniform_(-8, 8, generator=g))
    if seg:
        st = torch.zeros(M // seg, seg + 2, K, device=dev)
        st[:, 1:seg + 1] = a.view(M // seg, seg, K)
        opA = Operand(st, K, off=K, seg=seg, seg_stride=(seg + 2) * K)
    else:                                    # dG: rows of 4H floats, 3H of them read
        st = torch.zeros(M, K // 3 * 4, device=dev)
        st[:, :K] = a
        opA = Operand(st, K // 3 * 4)
    b = torch.randn((K, N) if bkm else (N, K), device=dev, generator=g) * 0.05
    bias = None if bkm else torch.randn(N, device=dev, generator=g)
    C = torch.zeros(M, N, device=dev)
    rows = slice(0, min(M, 8192))            # the float64 reference on a slice of the rows
    bt = (b if bkm else b.T).double()
    ref = a[rows].double() @ bt + (0 if bias is None else bias.double())
    mag = a[rows].double().abs() @ bt.abs() + (0 if bias is None else bias.double().abs())
    row = {}
    for name, split in (("f32", None), ("split2", 0), ("split1", 1)):
        call = lambda: ops.gemm(M, N, K, opA, 0, Operand(b, b.shape[1]), bkm, C, N, bias=bias, split=split)  # noqa: E731
        call()
        torch.cuda.synchronize()
        err = (C[rows].double() - ref).abs()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        call()
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        row[name] = (float((err / mag).max()), float(err.max() / ref.abs().max()), ms, 2.0 * M * N * K / ms / 1e9)
    return row


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    shapes = [  # M, N, K, n, sk, gap, seg, label
        (96, 136, 512, 3, 8, 0, 0, "check_gemm_group #1"), (64, 100, 640, 5, 16, 32, 0, "check_gemm_group #3 (gap)"),
        (200, 30, 512, 4, 8, 0, 0, "check_gemm_group #4"), (96, 48, 512, 2, 8, 0, 0, "check_gemm_group #5"),
        (768, 256, 8192, 6, 32, 256, 0, "check_gemm_group big"), (768, 256, 4096, 1, 8, 0, 0, "check_gemm_cases TN"),
        (768, 256, 122880, 6, 32, 256, 30, "configs[1] dW_hh x6 (dominant launch)"), (768, 512, 122880, 2, 48, 0, 30, "configs[1] layer-1 dW_ih x2"),
        (768, 256, 61440, 2, 48, 256, 15, "configs[1] future dW_hh x2"),
    ]
    if not a.quick:
        shapes += [(1536, 512, 491520, 6, 16, 512, 60, "configs[3] dW_hh x6"), (768, 256, 7680, 6, 8, 256, 30, "batch 256 dW_hh x6")]
    print(f"{'shape':44s} {'kernel':8s} {'max err/sum|a||b|':>18s} {'max err/max|C|':>15s} {'ms':>8s} {'TF(f32-eq)':>10s}")
    for (M, N, K, n, sk, gap, seg, label) in shapes:
        for wide in ((True, False) if K >= 100000 and N == 256 and n == 6 and M == 768 else (True,)):
            r = run(M, N, K, n, sk, gap, seg, a.reps, wide=wide)
            tag = f"{label} {M}x{N}x{K} x{n}" + ("" if wide else " N(0,1)")
            for name in ("f32", "split2", "split1"):
                if r[name] is None:
                    print(f"{tag:44s} {name:8s} not eligible")
                else:
                    e1, e2, ms, tf = r[name]
                    print(f"{tag:44s} {name:8s} {e1:18.3e} {e2:15.3e} {ms:8.3f} {tf:10.1f}")
    rows = [(122880, 768, 512, 30, 0, "configs[1] layer-1 projection gi = y W_ih^T + b"), (122880, 512, 768, 0, 1, "configs[1] layer-1 dY = dG W_ih"),
            (7680, 768, 512, 30, 0, "batch 256 layer-1 projection")]
    if not a.quick:
        rows += [(245760, 1536, 1024, 60, 0, "configs[3] layer-1 projection"), (245760, 1024, 1536, 0, 1, "configs[3] layer-1 dY")]
    for (M, N, K, seg, bkm, label) in rows:
        r = run_rows(M, N, K, seg, bkm, a.reps)
        tag = f"{label} {M}x{N}x{K}"
        for name in ("f32", "split2", "split1"):
            e1, e2, ms, tf = r[name]
            print(f"{tag:64s} {name:8s} {e1:18.3e} {e2:15.3e} {ms:8.3f} {tf:10.1f}")
    print("peak for frac: f32 MFMA 157.3 TF; bf16 dense 2500 TF / 6 products = 416.7 TF fp32-equivalent")


if __name__ == "__main__":
    main()
