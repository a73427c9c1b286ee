"""Timing-only ablations of the wave-specialised BPTT kernel (tuning build: make ab; VAME_LIB=tools/libvame_hip_ab.so)."""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from vame_amd import _lib
_lib._lib = _lib._bind(os.path.join(R, "tools", "libvame_hip_ab.so"))
import kernel_cases as kc
dev, H, B, T = "cuda", 256, 4096, 30
x, st, Y, hN = kc.run_gru_fwd(dev, H, B, T, seed=1)
dY = torch.randn(B, T, 2 * H, device=dev); dhN = torch.randn(B, 2 * H, device=dev)
def run(env, n=3):
    for k in ("VAME_GRU_WS", "VAME_WS_ABL", "VAME_ABL_BWD"): os.environ.pop(k, None)
    os.environ.update(env)
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); kc._run_gru_bwd(dev, H, B, T, st, Y, dY, dhN); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return ts
cfgs = [("lock-step", dict(VAME_GRU_WS="0"))] + [(n, dict(VAME_GRU_WS="1", VAME_WS_ABL=str(a))) for n, a in (
    ("ws full", 0), ("ws no copy-out", 1), ("ws no next loads", 2), ("ws no copy-out, no loads", 3), ("ws no MFMA loop", 4), ("ws no dgi_n stores", 8),
    ("ws no HBM at all (11)", 11), ("ws phase A + barriers only (7)", 7))]
for _ in range(10):                      # clocks / caches warm
    run(cfgs[0][1], 2); run(cfgs[1][1], 2)
acc = {n: [] for n, _ in cfgs}
for rnd in range(6):                     # interleaved rounds
    for n, e in cfgs:
        acc[n] += run(e, 3)
for n, _ in cfgs:
    v = sorted(acc[n]); print(f"{n:40s} median {v[len(v)//2]:.0f} us  min {v[0]:.0f}")
