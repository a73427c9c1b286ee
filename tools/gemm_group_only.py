#!/usr/bin/env python3
"""Run the grouped weight-gradient GEMM of the B=4096 step (6 x TN M=768 N=256 K=122880, split-K 32) a few times
(for rocprofv3 --pmc passes).  usage: gemm_group_only.py [count] [splitk] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vame_amd import ops
from vame_amd.ops import Operand
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
sk = int(sys.argv[2]) if len(sys.argv) > 2 else 32
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
M, N, K = 768, 256, 4096 * 30
A = [torch.randn(K, M, device="cuda") for _ in range(n)]
B = [torch.randn(K, N, device="cuda") for _ in range(n)]
C = torch.empty(n * M * N, device="cuda")
ws = torch.empty(n * sk * M * N, device="cuda")
for _ in range(reps):
    ops.gemm_group(M, N, K, [Operand(a, M) for a in A], 1, [Operand(b, N) for b in B], 1, C, [g * M * N for g in range(n)], N, sk, ws)
torch.cuda.synchronize()
