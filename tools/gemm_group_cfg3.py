#!/usr/bin/env python3
"""The dominant grouped weight-gradient launch of BASELINE configs[3] (6 x TN M=1536 N=512 K=491,520: the dW_hh of the H = 512, T = 60, batch 8192 step) at
several split-K factors: time per launch (HIP events) -- and, under `rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE` (tools/traffic_gemm_cfg3.sh), the HBM
traffic per launch, which is told apart by the grid size.   usage: gemm_group_cfg3.py [split-K ...]   (default 8 16 32 64)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vame_amd import ops  # noqa: E402
from vame_amd.ops import Operand  # noqa: E402

sks = [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64]
n, H, T, Bt = 6, 512, 60, 8192
M, N, K = 3 * H, H, Bt * T
A = [torch.randn(K, 4 * H, device="cuda") for _ in range(n)]                     # dG (B, T, 4H): [da_r | da_z | dgi_n | dgh_n]
Y = [torch.randn(Bt, T + 2, 2 * H, device="cuda") for _ in range(n)]             # the (B, T + 2, 2H) sequences: h_{t-1} = a strided view
C = torch.empty(n * M * N, device="cuda")
ws = torch.empty(n * max(sks) * M * N, device="cuda")
opA = [Operand(a, 4 * H) for a in A]
opB = [Operand(y, 2 * H, off=0, seg=T, seg_stride=(T + 2) * 2 * H) for y in Y]
alg = n * K * (M + N) * 4
for sk in sks:
    f = lambda: ops.gemm_group(M, N, K, opA, 1, opB, 1, C, [g * M * N for g in range(n)], N, sk, ws, a_gap_at=2 * H, a_gap=H)  # noqa: E731
    f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    tiles = (M // 128) * (N // 128) * n
    print(f"split-K {sk:3d}: grid {tiles * sk * 256:9d} threads  {ms:8.3f} ms  {2.0 * M * N * K * n / ms / 1e9:6.1f} TF   operands {alg / 1e9:.2f} GB + partial sums w+r {2 * n * sk * M * N * 4 / 1e9:.2f} GB")
