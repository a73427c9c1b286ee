#!/usr/bin/env python3
"""Per-step GEMMs of the step-wise BPTT path at BASELINE config 4 (H=512, batch 8192): split-K choice for dh += dgh W_hh (NN, M=8192 N=512 K=1536)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import torch
import microbench as mb
from vame_amd import ops
from vame_amd.ops import Operand
for (M, N, K, akm, bkm) in [(8192, 512, 1536, 0, 1), (8192, 1536, 512, 0, 0)]:
    A = torch.randn((K, M) if akm else (M, K), device="cuda")
    B = torch.randn((K, N) if bkm else (N, K), device="cuda")
    C = torch.zeros(M, N, device="cuda")
    for sk in (1, 2, 3, 4):
        ws = torch.empty(sk * M * N, device="cuda") if sk > 1 else None
        ms = mb.timeit(lambda: ops.gemm(M, N, K, Operand(A, A.shape[1]), akm, Operand(B, B.shape[1]), bkm, C, N, accumulate=True, splitk=sk, ws=ws), reps=20)
        print(f"M={M} N={N} K={K} akm={akm} bkm={bkm} sk={sk}: {ms*1e3:7.1f} us {2.0*M*N*K/ms/1e9:6.1f} TF", flush=True)
