#!/usr/bin/env python3
"""Shader clock actually sustained inside the GRU sequence kernels (tuning build: make probe).
Every workgroup stamps s_memtime (shader-clock counter) and s_memrealtime (100 MHz) at entry and exit;
ratio x 100 MHz = average clock over the kernel.  usage: VAME_LIB=tools/libvame_hip_probe.so python tools/probe_clock.py"""
import ctypes
import os
import sys

sys.argv = sys.argv[:1] + ["5"]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402
import microbench as mb  # noqa: E402
from vame_amd import _lib  # noqa: E402

L = _lib._lib
L.vame_probe_set_gru.argtypes = [ctypes.c_void_p]
probe = torch.zeros((1 << 16) + (1 << 14) * 24 + (1 << 14) * 64, dtype=torch.int64, device="cuda")
L.vame_probe_set_gru(probe.data_ptr())


FWD_PHASES = ["loop-top", "mfma", "gates+h->lds", "stash-st", "barrier", "y-store", "", "loop-groups"]
BWD_PHASES = ["loop-top", "coef+lds", "barrier1", "dG-copy", "ld-issue", "mfma", "barrier2", "loop-groups"]
T_CUR = [30]


def report(tag, nwg):
    torch.cuda.synchronize()
    p = probe[:4 * nwg].view(-1, 4).cpu().numpy().astype("float64")
    ph = probe[1 << 16:(1 << 16) + 24 * nwg].view(-1, 24).cpu().numpy().astype("float64")[p[:, 1] > 0]
    p = p[p[:, 1] > 0]
    span = (p[:, 3].max() - p[:, 2].min()) / 100.0
    print(f"{tag}: {len(p)} workgroups, clock ratio {p[:, 0].sum() / p[:, 1].sum():.3f} (x100 MHz), workgroup length "
          f"min {p[:, 1].min() / 100:.1f} p50 {sorted(p[:, 1])[len(p) // 2] / 100:.1f} max {p[:, 1].max() / 100:.1f} us, span {span:.1f} us", flush=True)
    tot = ph[:, :8].sum()
    print("      MFMA-loop groups (cycles/step): " + " ".join(f"{ph[:, 8 + g].mean() / T_CUR[0]:.0f}" for g in range(16) if ph[:, 8 + g].sum() > 0), flush=True)
    names = FWD_PHASES if "fwd" in tag else BWD_PHASES
    print("      wave-0 cycles per step: " + "  ".join(f"{n} {ph[:, i].mean() / T_CUR[0]:.0f}" for i, n in enumerate(names[:7]) if n)
          + f"   (sum {tot / len(ph) / T_CUR[0]:.0f}; MFMA floor 2 waves x 384 x 64 = 49152)", flush=True)
    pw = probe[(1 << 16) + (1 << 14) * 24:(1 << 16) + (1 << 14) * 24 + 64 * nwg].view(-1, 8, 8).cpu().numpy().astype("float64")
    pw = pw[pw[:, 0, :].sum(1) > 0]
    if len(pw):
        for i, n in enumerate(names):
            if n:
                print(f"      per wave, {n:>12s}: " + " ".join(f"{pw[:, w_, i].mean() / T_CUR[0]:7.0f}" for w_ in range(8)), flush=True)
    probe.zero_()


for (H, B, T, ns) in ((256, 4096, 30, 2), (256, 4096, 30, 4), (256, 256, 30, 2)):
    mb.bench_gru(H, B, T, ns, hook=lambda which: report(f"  gru_{which} H={H} B={B} T={T} streams={ns}", ns * ((B + 31) // 32) + 64))
