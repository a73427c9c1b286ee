import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from vame_amd import _lib, ops
if os.environ.get("VAME_LIB"):
    _lib._lib = _lib._bind(os.environ["VAME_LIB"])
torch.manual_seed(0)
Z, B = 30, 4096
vstate = torch.zeros(32 * 32, device="cuda", dtype=torch.float64)
losses, Minv, G = torch.zeros(8, device="cuda"), torch.zeros(Z, Z, device="cuda"), torch.zeros(Z, Z, device="cuda")
z = torch.randn(B, Z, device="cuda")
ref = None
ts = []
for it in range(40):
    z += 0.01 * torch.randn_like(z)            # slowly drifting latents, like consecutive optimizer steps
    G.copy_(z.t() @ z)
    losses.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.nuclear(G, Z, Z, B, 0.1, float(B), losses, 3, Minv, gscale=1.0, vstate=vstate)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
    sv = torch.linalg.svdvals((z.double() / B ** 0.5))
    exact = 0.1 * float(sv.sum())
print(f"nuclear kernel: median {sorted(ts)[len(ts)//2]:.1f} us (first {ts[0]:.1f}); loss {float(losses[3]):.7f} vs svd {exact:.7f} rel {abs(float(losses[3])-exact)/exact:.2e}")
