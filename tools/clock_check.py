"""Validation of vame_clock_stamp / ops.ClockProbe on the MI355X: clock over an idle stretch, over back-to-back MFMA-bound GEMMs of
different lengths, and how far the s_memtime counters of different compute units are offset against each other."""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import numpy as np, torch
from vame_amd import ops
from vame_amd.ops import Operand
dev = torch.device("cuda")
M, N, K = 768, 512, 122880
A, B = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
C, ws = torch.zeros(M, N, device=dev), torch.zeros(64 * M * N, device=dev)
def gemm(n):
    for _ in range(n):
        ops.gemm(M, N, K, Operand(A, M), 1, Operand(B, N), 1, C, N, splitk=32, ws=ws)
gemm(20); torch.cuda.synchronize()
p = ops.ClockProbe(dev)
p.start(); p.stop(); torch.cuda.synchronize()
b = p.buf.cpu().numpy()
for k in (0, 1):
    ratio = b[k][:, 0] - 24 * b[k][:, 1]
    print(f"stamp {k}: {len(set((int(x), int(h) & 0xFF00) for _, _, x, h in b[k]))} distinct CUs; memtime - 24 x realtime spread over CUs: "
          f"{int(ratio.max() - ratio.min())} ticks; realtime spread {int(b[k][:, 1].max() - b[k][:, 1].min())} ticks (10 ns each)")
for label, fn in (("idle 2 ms", lambda: torch.cuda._sleep(int(2e-3 * 2.4e9))), ("1 GEMM (~0.9 ms)", lambda: gemm(1)), ("10 GEMMs", lambda: gemm(10)), ("100 GEMMs", lambda: gemm(100)),
                  ("tiny kernel", lambda: C.zero_())):
    for rep in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        p.start(); e0.record(); fn(); e1.record(); p.stop(); torch.cuda.synchronize()
        print(f"{label:18s} {e0.elapsed_time(e1) * 1e3:10.1f} us  clock {p.mhz()} MHz")
