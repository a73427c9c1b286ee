"""The second encoder layer's input projection gi = Y0 W_ih^T + b (M = B T, N = 3H, K = 2H) with the weight operand as stored ((N, K) row-major: NT)
against a k-major copy of it ((K, N): NN) -- interleaved HIP-event medians.   python tools/proj_layout_ab.py [B=4096] [T=30] [H=256]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vame_amd import ops
from vame_amd.ops import Operand
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 30
H = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = "cuda"
M, N, K = B * T, 3 * H, 2 * H
Y = torch.randn(B, T + 2, K, device=dev)
W = torch.randn(N, K, device=dev) / K ** 0.5
Wt = W.t().contiguous()
bias = torch.randn(N, device=dev)
gi1, gi2 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
A = Operand(Y, K, off=K, seg=T, seg_stride=(T + 2) * K)
runs = dict(NT=lambda: ops.gemm(M, N, K, A, 0, Operand(W, K), 0, gi1, N, bias=bias), NN=lambda: ops.gemm(M, N, K, A, 0, Operand(Wt, N), 1, gi2, N, bias=bias))
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
t = {k: [] for k in runs}
for _ in range(7):
    for k, fn in runs.items(): t[k].append(timeit(fn))
for k, v in t.items():
    med = sorted(v)[len(v) // 2]
    print(f"{k}: {med:8.1f} us  {2.0 * M * N * K / med / 1e6:6.1f} TF")
print("max |NT - NN| =", float((gi1 - gi2).abs().max()), " bit-identical:", bool(torch.equal(gi1, gi2)))
