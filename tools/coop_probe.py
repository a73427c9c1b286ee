"""Per-wave phase cycles of the cooperative GRU kernels (probe build: make probe; loads tools/libvame_hip_probe.so).
forward phases per step: 0 acc init + MFMA loop, 1 K-quarter exchange, 2 gate math, 3 publish (packets, sequence, stash) issue,
4 (nothing), 5 packet poll, 6 LDS rebuild, 7 barrier.   BPTT: 0 element-wise + A tile -> LDS, 1 barrier, 2 dG stores + stash
prefetch issue, 3 MFMA loop, 4 publish + drain, 5 barrier, 6 flag + poll + barrier, 7 reduce-scatter loads + sums."""
import ctypes, os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import torch
from vame_amd import _lib, ops
_lib._lib = _lib._bind(os.path.join(R, "tools", "libvame_hip_probe.so"))
import microbench as mb
L = _lib.lib()
L.vame_probe_set_coop.argtypes = [ctypes.c_void_p]
NWV = 8                                            # waves per workgroup (COOP_NT / 64)
probe = torch.zeros(256 * NWV * 8, dtype=torch.int64, device="cuda")
L.vame_probe_set_coop(probe.data_ptr())
state = ops.CoopState(torch.device("cuda"))
orig, orig_b = ops.gru_seq_fwd, ops.gru_seq_bwd
H, T = 256, 30
FN = ["mfma", "k-xchg", "gates", "publish", "barrier", "packet poll", "lds rebuild", "barrier"]
BN = ["elementwise", "barrier", "dG+prefetch", "mfma", "publish+drain", "barrier", "flag/poll", "reduce"]
for (B, ns) in ((256, 2), (256, 4)):
    for name, kern in (("auto", ops.KERNEL_AUTO), ("32-row groups", ops.KERNEL_LOCKSTEP)):
        if ns == 4 and kern == ops.KERNEL_LOCKSTEP:
            continue
        res = {}
        def fwd(rows, B_, H_, *a, **k):
            ops.gru_coop_fwd(rows, B_, H_, state, kernel=kern)
            res["f"] = probe.view(-1, NWV, 8).cpu().numpy().astype("float64"); probe.zero_()
        def bwd(rows, B_, H_, *a, **k):
            ops.gru_coop_bwd(rows, B_, H_, state, kernel=kern)
            res["b"] = probe.view(-1, NWV, 8).cpu().numpy().astype("float64"); probe.zero_()
        ops.gru_seq_fwd, ops.gru_seq_bwd = fwd, bwd
        probe.zero_()
        f1, b1 = mb.bench_gru(H, B, T, ns, quiet=True)
        ops.gru_seq_fwd, ops.gru_seq_bwd = orig, orig_b
        print(f"B={B} streams={ns} {name}: (timings include the probe copies; cycles per step, mean over workgroups, waves 0..7)")
        for tag, names in (("f", FN), ("b", BN)):
            p = res[tag]
            p = p[p.sum((1, 2)) > 0]
            print(f"  {'fwd' if tag == 'f' else 'bwd'}: {len(p)} workgroups, total per step " + " ".join(f"{p[:, w_, :].sum(1).mean() / T:7.0f}" for w_ in range(NWV)))
            for i, n in enumerate(names):
                print(f"      {n:>14s}: " + " ".join(f"{p[:, w_, i].mean() / T:7.0f}" for w_ in range(NWV)))
