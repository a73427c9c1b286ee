#!/usr/bin/env python3
"""Per-wave phase times (shader-clock cycles per step) of the BPTT kernels, lock-step vs wave-specialised (probe build: make probe).
Waves 0-3 of the wave-specialised kernel are the MFMA waves, 4-7 the memory waves; phase names follow the lock-step kernel
(for the MFMA waves `coef+lds` = carry read, `ld-issue` = d -> xd + dy request)."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["VAME_LIB"] = os.path.join(R, "tools", "libvame_hip_probe.so")
sys.argv = sys.argv[:1] + ["5"]
sys.path.insert(0, os.path.join(R, "tools"))
import torch
import microbench as mb
from vame_amd import _lib
L = _lib._lib
L.vame_probe_set_gru.argtypes = [ctypes.c_void_p]
probe = torch.zeros((1 << 16) + (1 << 14) * 24 + (1 << 14) * 64, dtype=torch.int64, device="cuda")
L.vame_probe_set_gru(probe.data_ptr())
NAMES = ["loop-top", "coef+lds", "barrier1", "dG-copy", "ld-issue", "mfma", "barrier2", "loop-groups"]
T = 30


def report(tag, nwg):
    torch.cuda.synchronize()
    p = probe[:4 * nwg].view(-1, 4).cpu().numpy().astype("float64")
    p = p[p[:, 1] > 0]
    print(f"{tag}: {len(p)} workgroups, clock {p[:, 0].sum() / p[:, 1].sum() * 100:.0f} MHz, workgroup length p50 {sorted(p[:, 1])[len(p) // 2] / 100:.1f} us", flush=True)
    pw = probe[(1 << 16) + (1 << 14) * 24:(1 << 16) + (1 << 14) * 24 + 64 * nwg].view(-1, 8, 8).cpu().numpy().astype("float64")
    pw = pw[pw[:, :, :].sum((1, 2)) > 0]
    for i, n in enumerate(NAMES[:7]):
        print(f"      per wave, {n:>10s}: " + " ".join(f"{pw[:, w_, i].mean() / T:7.0f}" for w_ in range(8)), flush=True)
    print(f"      per wave, {'sum':>10s}: " + " ".join(f"{pw[:, w_, :7].sum(1).mean() / T:7.0f}" for w_ in range(8)), flush=True)
    probe.zero_()


combos = [("0", 0, 0), ("1", 0, 0)] + [("1", int(a), int(b)) for a, b in (x.split(":") for x in os.environ.get("WS_SWEEP", "").split(",") if x)]
for ws, cp, ld in combos:
    os.environ.update(VAME_GRU_WS=ws, VAME_WS_PACE_CP=str(cp), VAME_WS_PACE_LD=str(ld))
    mb.bench_gru(256, 4096, T, 2, quiet=True, hook=lambda which: report(f"gru_{which} WS={ws} pace copy {cp} loads {ld}", 2 * 128 + 64) if which == "bwd" else probe.zero_())
