#!/usr/bin/env python3
"""Does the two-level (batch, time) row addressing cost anything in the large NT / NN / TN GEMMs of the step?  Plain operands vs the
padded-sequence operands the engine passes (seg = T rows of a (T+2)-slot record), production library."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from vame_amd import ops
from vame_amd.ops import Operand
B, T, H = 4096, 30, 256
def timeit(f, reps=8):
    for _ in range(3): f()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
Ypad = torch.randn(B, T + 2, 2 * H, device="cuda"); Yflat = torch.randn(B * T, 2 * H, device="cuda")
W = torch.randn(3 * H, 2 * H, device="cuda"); gi = torch.empty(B * T, 3 * H, device="cuda"); bias = torch.randn(3 * H, device="cuda")
M, N, K = B * T, 3 * H, 2 * H
fl = 2.0 * M * N * K
for name, A in (("NT gi projection, plain A", Operand(Yflat, 2 * H)), ("NT gi projection, padded-sequence A", Operand(Ypad, 2 * H, off=2 * H, seg=T, seg_stride=(T + 2) * 2 * H))):
    ms = timeit(lambda: ops.gemm(M, N, K, A, 0, Operand(W, 2 * H), 0, gi, N, bias=bias))
    print(f"{name:48s} {ms*1e3:8.1f} us {fl/ms/1e9:6.1f} TF")
# TN weight gradient dW_hh: A = dG (gap), B = h_{t-1} rows (padded sequence) vs plain
dG = torch.randn(B * T, 4 * H, device="cuda"); hp_flat = torch.randn(B * T, H, device="cuda")
Cw = torch.empty(3 * H * H, device="cuda"); ws = torch.empty(32 * 3 * H * H, device="cuda")
M2, N2, K2 = 3 * H, H, B * T
fl2 = 2.0 * M2 * N2 * K2
for name, Bop, gap in (("TN dW_hh, plain B, no gap", Operand(hp_flat, H), (0, 0)), ("TN dW_hh, plain B, column gap", Operand(hp_flat, H), (2 * H, H)),
                       ("TN dW_hh, padded-sequence B, column gap", Operand(Ypad, 2 * H, off=0, seg=T, seg_stride=(T + 2) * 2 * H), (2 * H, H))):
    ms = timeit(lambda: ops.gemm(M2, N2, K2, Operand(dG, 4 * H), 1, Bop, 1, Cw, N2, splitk=96, ws=torch.empty(96 * M2 * N2, device="cuda") if False else ws.new_empty(96 * M2 * N2), a_gap_at=gap[0], a_gap=gap[1]))
    print(f"{name:48s} {ms*1e3:8.1f} us {fl2/ms/1e9:6.1f} TF")
