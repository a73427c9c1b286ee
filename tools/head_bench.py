#!/usr/bin/env python3
"""Output head of a decoder in the training step: the fused kernel (vame_head_fused_f32) against the three launches it replaces
(hidden_to_output GEMM, MSE kernel, dY GEMM).  HIP events, batch 4096.  usage: python tools/head_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vame_amd import ops  # noqa: E402
from vame_amd.ops import Operand  # noqa: E402

dev = "cuda"


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (B, T, F, K) in ((4096, 30, 24, 512), (4096, 15, 24, 512), (8192, 60, 24, 1024)):
    Y = torch.randn(B, T + 2, K, device=dev)
    W = torch.randn(F, K, device=dev) / K ** 0.5
    bias = torch.randn(F, device=dev)
    row = (T + 15) * F
    win = torch.randn(B, row, device=dev)
    pred, dpred = torch.empty(B * T, F, device=dev), torch.empty(B * T, F, device=dev)
    dY = torch.empty(B * T, K, device=dev)
    loss = torch.zeros(4, device=dev)
    Yop = Operand(Y, K, off=K, seg=T, seg_stride=(T + 2) * K)
    fused = timeit(lambda: ops.head_fused(Yop, B * T, F, K, Operand(W, K), bias, win, 0, row, 2.0, pred, dpred, dY, K, loss, 0))
    t1 = timeit(lambda: ops.gemm(B * T, F, K, Yop, 0, Operand(W, K), 0, pred, F, bias=bias))
    t2 = timeit(lambda: ops.mse_fwd_bwd(pred, win, 0, row, B, T * F, 2.0, dpred, loss, 0))
    t3 = timeit(lambda: ops.gemm(B * T, K, F, Operand(dpred, F), 0, Operand(W, K), 1, dY, K))
    gb = (2 * B * T * K + 3 * B * T * F) * 4 / 1e9
    print(f"B={B} T={T} F={F} K={K}: fused {fused:7.1f} us ({gb / fused * 1e3:5.2f} TB/s)   separate {t1:6.1f} + {t2:5.1f} + {t3:6.1f} = {t1 + t2 + t3:7.1f} us")
