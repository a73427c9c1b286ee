#!/usr/bin/env python3
"""Run one GEMM shape a few times (for rocprofv3 --pmc runs).  usage: gemm_only.py M N K akm bkm sk [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vame_amd import ops
from vame_amd.ops import Operand
M, N, K, akm, bkm, sk = [int(v) for v in sys.argv[1:7]]
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 5
dev = "cuda"
A = torch.randn((K, M) if akm else (M, K), device=dev)
B = torch.randn((K, N) if bkm else (N, K), device=dev)
C = torch.empty(M, N, device=dev)
ws = torch.empty(sk * M * N, device=dev) if sk > 1 else None
for _ in range(reps):
    ops.gemm(M, N, K, Operand(A, A.shape[1]), akm, Operand(B, B.shape[1]), bkm, C, N, splitk=sk, ws=ws)
torch.cuda.synchronize()
