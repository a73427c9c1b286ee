"""Interleaved A/B of engine attributes on the train step (HIP-event medians): python tools/step_ab.py B name=attr:val,attr:val name2=...
e.g.  python tools/step_ab.py 256 join=nuc_join_before_coop:1 nojoin=nuc_join_before_coop:0 inline=nuc_side:0"""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import numpy as np, torch
import bench
from vame_amd.model.dataloader import DeviceWindowLoader
from vame_amd.model.rnn_model import RNN_VAE
from vame_amd.model.rnn_vae import FusedAdamAMSGrad
B = int(sys.argv[1])
variants = []
for a in sys.argv[2:]:
    name, spec = a.split("=")
    variants.append((name, [(kv.split(":")[0], int(kv.split(":")[1])) for kv in spec.split(",") if kv]))
H, T = int(os.environ.get("AB_H", "256")), int(os.environ.get("AB_T", "30"))
dev = torch.device("cuda")
torch.manual_seed(19)
model = RNN_VAE(2 * T, bench.Z, bench.F, 1, bench.FS, H, H, H, H, 0, 0, 0, False).to(dev).train()
opt = FusedAdamAMSGrad(model, lr=5e-4)
loader = DeviceWindowLoader(bench._SynthDataset(T), B, T + bench.FS, dev, rank=0, world=1)
eng = model._ensure_engine()
base = {k: getattr(eng, k) for _, kv in variants for k, _ in kv}
def step():
    win = loader.gather(loader.draw_starts())
    model.loss_step(win, 1.0, beta=1.0, kloss=bench.Z, klmbda=0.1, bsize=B)
    opt.step()
def run(kv, n):
    for k, v in base.items(): setattr(eng, k, v)
    for k, v in kv: setattr(eng, k, type(base[k])(v))
    step(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): step()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
n = int(os.environ.get("AB_STEPS", "20"))
for _ in range(2):
    for _, kv in variants: run(kv, 5)
res = {name: [] for name, _ in variants}
for _ in range(int(os.environ.get("AB_ROUNDS", "7"))):
    for name, kv in variants: res[name].append(run(kv, n))
for name in res:
    v = sorted(res[name])
    print(f"B={B} {name:12s}: median {v[len(v) // 2]:7.3f} ms/step  (min {v[0]:.3f} max {v[-1]:.3f})  {B / v[len(v) // 2]:8.1f} k windows/s")
