"""Per-wave phase cycles of the streaming output-head kernel (probe build: make probe; loads tools/libvame_hip_probe.so).
phases per tile: 0 tile -> LDS (waits for the prefetch) + requests + barrier, 1 P1 + barrier, 2 dpred phase + barrier, 3 prefetch issue + P3 + barrier,
4 P2 + barrier, 5 copy-out.   usage: python tools/head_probe.py [B=4096] [T=30] [K=512]"""
import ctypes, os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import torch
from vame_amd import _lib, ops
from vame_amd.ops import Operand
_lib._lib = _lib._bind(os.path.join(R, "tools", "libvame_hip_probe.so"))
L = _lib.lib()
L.vame_probe_set_head.argtypes = [ctypes.c_void_p]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 30
K = int(sys.argv[3]) if len(sys.argv) > 3 else 512
F, dev = 24, "cuda"
probe = torch.zeros(1024 * 4 * 8, dtype=torch.int64, device=dev)
L.vame_probe_set_head(probe.data_ptr())
Y = torch.randn(B, T + 2, K, device=dev)
W = torch.randn(F, K, device=dev) / K ** 0.5
bias = torch.randn(F, device=dev)
row = (T + 15) * F
win = torch.randn(B, row, device=dev)
pred, dpred = torch.empty(B * T, F, device=dev), torch.empty(B * T, F, device=dev)
dY = torch.empty(B * T, K, device=dev)
dW = torch.empty(F * K, device=dev)
loss = torch.zeros(4, device=dev)
ws = torch.empty(ops.head_stream_ws_floats(B * T, F, K), device=dev)
Yop = Operand(Y, K, off=K, seg=T, seg_stride=(T + 2) * K)
run = lambda: ops.head_stream(Yop, B * T, F, K, Operand(W, K), bias, win, 0, row, 2.0, pred, dpred, dY, K, loss, 0, dW, 0, ws)
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
p = probe.view(-1, 4, 8).cpu().numpy().astype("float64")
p = p[p.sum((1, 2)) > 0]
ntiles = -(-B * T // 16)
per = ntiles / len(p)
names = ["tile->LDS + wait", "P1", "dpred", "P3 (+prefetch issue)", "P2", "copy-out"]
print(f"B={B} T={T} K={K}: {e0.elapsed_time(e1) * 1e3:.1f} us (with probes), {len(p)} workgroups, {per:.1f} tiles each; cycles per tile, mean over workgroups, waves 0..3")
print(f"  total: " + " ".join(f"{p[:, w, :].sum(1).mean() / per:8.0f}" for w in range(4)))
for i, n in enumerate(names):
    print(f"  {n:>22s}: " + " ".join(f"{p[:, w, i].mean() / per:8.0f}" for w in range(4)))
