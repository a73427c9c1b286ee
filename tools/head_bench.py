#!/usr/bin/env python3
"""Output head of a decoder in the training step: the streaming kernel (vame_head_stream_f32: one pass over the states) against the launches it
replaces (hidden_to_output GEMM, MSE kernel, dY GEMM, split-K weight-gradient GEMM).  HIP events, interleaved medians.
usage: python tools/head_bench.py [B ...]"""
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


batches = [int(a) for a in sys.argv[1:]] or [4096, 256]
for B in batches:
    for (T, F, K) in ((30, 24, 512), (15, 24, 512), (30, 24, 256)):
        Y = torch.randn(B, T + 2, K, device=dev)
        W = torch.randn(F, K, device=dev) / K ** 0.5
        bias = torch.randn(F, device=dev)
        row = (T + 15) * F
        win = torch.randn(B, row, device=dev)
        pred, dpred = torch.empty(B * T, F, device=dev), torch.empty(B * T, F, device=dev)
        dY = torch.empty(B * T, K, device=dev)
        dW = torch.empty(F * K, device=dev)
        loss = torch.zeros(4, device=dev)
        Yop = Operand(Y, K, off=K, seg=T, seg_stride=(T + 2) * K)
        ws = torch.empty(ops.head_stream_ws_floats(B * T, F, K), device=dev)
        sk = max(8, min(96, (B * T) // 2048 // 8 * 8))
        wsk = torch.empty(sk * F * K, device=dev)
        runs = dict(
            stream=lambda: ops.head_stream(Yop, B * T, F, K, Operand(W, K), bias, win, 0, row, 2.0, pred, dpred, dY, K, loss, 0, dW, 0, ws),
            pred=lambda: ops.gemm(B * T, F, K, Yop, 0, Operand(W, K), 0, pred, F, bias=bias),
            mse=lambda: ops.mse_fwd_bwd(pred, win, 0, row, B, T * F, 2.0, dpred, loss, 0),
            dY=lambda: ops.gemm(B * T, K, F, Operand(dpred, F), 0, Operand(W, K), 1, dY, K),
            dW=lambda: ops.gemm(F, K, B * T, Operand(dpred, F), 1, Yop, 1, dW, K, splitk=sk, ws=wsk))
        t = {k: [] for k in runs}
        for _ in range(5):
            for k, fn in runs.items():
                t[k].append(timeit(fn))
        m = {k: sorted(v)[len(v) // 2] for k, v in t.items()}
        gb = (2 * B * T * K + 3 * B * T * F) * 4 / 1e9
        sep = m["pred"] + m["mse"] + m["dY"] + m["dW"]
        print(f"B={B} T={T} F={F} K={K}: streaming head {m['stream']:7.1f} us ({gb / m['stream'] * 1e3:5.2f} TB/s of {gb * 1e3:.0f} MB)   separate "
              f"{m['pred']:6.1f} + {m['mse']:5.1f} + {m['dY']:6.1f} + {m['dW']:6.1f} = {sep:7.1f} us")
