#!/usr/bin/env python3
"""A/B of tile shapes / split-K for the dominant grouped weight-gradient launch (6 x TN M=768 N=256 K=122880); tuning build (make ab)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from vame_amd import _lib, ops
from vame_amd.ops import Operand
_lib._lib = _lib._bind(os.path.join(R, "tools", "libvame_hip_ab.so"))
n, M, N, K = 6, 768, 256, 4096 * 30
A = [torch.randn(K, 1024, device="cuda") for _ in range(n)]
B = [torch.randn(K, N, device="cuda") for _ in range(n)]
C = torch.empty(n * M * N, device="cuda")
ws = torch.empty(n * 128 * M * N, device="cuda")
def run(tile, var, sk, reps=4):
    os.environ.pop("VAME_GEMM_TILE", None); os.environ.pop("VAME_GEMM_VAR", None)
    if tile: os.environ["VAME_GEMM_TILE"] = str(tile)
    if var is not None: os.environ["VAME_GEMM_VAR"] = str(var)
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.gemm_group(M, N, K, [Operand(a, 1024) for a in A], 1, [Operand(b, N) for b in B], 1, C, [g * M * N for g in range(n)], N, sk, ws, a_gap_at=512, a_gap=256)
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return ts
cfgs = [("128x128 var5 sk32 (production)", 0, None, 32), ("128x128 var5 sk64", 0, None, 64), ("128x128 var13 sk32", 0, 13, 32),
        ("256x128 4w sk32", 1, None, 32), ("256x128 4w sk24", 1, None, 24), ("256x128 4w sk16", 1, None, 16), ("256x128 4w sk48", 1, None, 48),
        ("256x128 8w sk32", 2, None, 32), ("256x128 8w sk16", 2, None, 16), ("256x128 4w var13 sk32", 1, 13, 32)]
for _ in range(6): run(0, None, 32, 2)
acc = {c[0]: [] for c in cfgs}
for rnd in range(5):
    for name, tile, var, sk in cfgs:
        acc[name] += run(tile, var, sk, 3)
fl = 2.0 * M * N * K * n
for name, *_ in cfgs:
    v = sorted(acc[name]); med = v[len(v) // 2]
    print(f"{name:34s} median {med*1e3:8.1f} us  {fl/med/1e9:6.1f} TF   min {v[0]*1e3:8.1f}")
