"""Per-wave phase cycles of the forward kernels (probe build: make probe; VAME_LIB=tools/libvame_hip_probe.so): lock-step vs skewed.
Phases of the skewed kernel per step: 0 y copy-out issue, 1 part 0 (acc init + K loop over k < H/2), 4 barrier behind part 0, 2 part 1 K loop
(+ gi request), 3 gate math + stash stores, 5 barrier behind part 1."""
import ctypes, os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import torch
from vame_amd import _lib, ops
_lib._lib = _lib._bind(os.path.join(R, "tools", "libvame_hip_probe.so"))      # the probe build (make probe)
import fwd_table as ft
L = _lib.lib()
L.vame_probe_set_gru.argtypes = [ctypes.c_void_p]
probe = torch.zeros((1 << 16) + (1 << 14) * 24 + (1 << 14) * 64, dtype=torch.int64, device="cuda")
L.vame_probe_set_gru(probe.data_ptr())
H, B, T = 256, ft.B, ft.T
for form in os.environ.get("FORMS", "gi,xin,dec").split(","):
    rows, flops, keep = ft.rows_for(form, H)
    nwg = len(rows) * ((B + 31) // 32) + 64
    for name, kern, prio in (("lock-step", "1", "1"), ("skewed prio 1", "3", "1"), ("skewed prio 0", "3", "0")):
        os.environ.update(VAME_GRU_FWD=kern, VAME_GRU_FWD_PRIO=prio)
        for _ in range(3):
            ops.gru_seq_fwd(rows, B, H)
        torch.cuda.synchronize(); probe.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.gru_seq_fwd(rows, B, H); e1.record(); torch.cuda.synchronize()
        p = probe[:4 * nwg].view(-1, 4).cpu().numpy().astype("float64")
        p = p[p[:, 1] > 0]
        pw = probe[(1 << 16) + (1 << 14) * 24:(1 << 16) + (1 << 14) * 24 + 64 * nwg].view(-1, 8, 8).cpu().numpy().astype("float64")
        pw = pw[pw[:, 0, :].sum(1) > 0]
        steps = T if form != "dec" else None
        print(f"{form} {name}: {e0.elapsed_time(e1) * 1e3:.0f} us, clock ratio {p[:, 0].sum() / p[:, 1].sum():.2f}; workgroup cycles p50 {sorted(p[:, 0])[len(p) // 2]:.0f}")
        for i, n in enumerate(["y-copy", "part0/mfma", "part1", "gates", "bar-mid", "bar-end", "6"]):
            if pw[:, :, i].sum() > 0:
                print(f"      {n:>11s}: " + " ".join(f"{pw[:, w_, i].mean() / (steps or T):7.0f}" for w_ in range(8)))
    del rows, keep
