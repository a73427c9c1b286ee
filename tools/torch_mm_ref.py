#!/usr/bin/env python3
"""Vendor-library reference for the step's GEMM shapes: torch.mm / torch.matmul (rocBLAS / hipBLASLt fp32, TF32 off)
timed beside vame_gemm_f32 on the same operands.  usage: python tools/torch_mm_ref.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vame_amd import _lib, ops
if os.environ.get("VAME_LIB"):
    _lib._lib = _lib._bind(os.environ["VAME_LIB"])
from vame_amd.ops import Operand

torch.backends.cuda.matmul.allow_tf32 = False
dev = "cuda"


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


BT = 4096 * 30
for (M, N, K, akm, bkm, sk, label) in [(768, 256, BT, 1, 1, 64, "dW_hh TN"), (768, 512, BT, 1, 1, 32, "dW_ih TN"),
                                       (BT, 768, 512, 0, 0, 1, "gi NT"), (BT, 512, 768, 0, 1, 1, "dx NN")]:
    A = torch.randn((K, M) if akm else (M, K), device=dev)
    B = torch.randn((K, N) if bkm else (N, K), device=dev)
    C = torch.empty(M, N, device=dev)
    ws = torch.empty(sk * M * N, device=dev) if sk > 1 else None
    mine = timeit(lambda: ops.gemm(M, N, K, Operand(A, A.shape[1]), akm, Operand(B, B.shape[1]), bkm, C, N, splitk=sk, ws=ws))
    At = A.t() if akm else A
    Bt = B if bkm else B.t()
    ref = timeit(lambda: torch.mm(At, Bt, out=C))
    fl = 2.0 * M * N * K / 1e9
    print(f"{label:10s} M={M} N={N} K={K}: vame {mine*1e3:8.1f} us {fl/mine:6.1f} TF | torch.mm {ref*1e3:8.1f} us {fl/ref:6.1f} TF", flush=True)
