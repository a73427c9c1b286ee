"""Host enqueue time vs total time of one fused train step (forward + loss + backward) at a given batch size.
usage: python tools/host_time.py [batch]"""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from vame_amd.model.rnn_model import RNN_VAE
torch.manual_seed(19)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = RNN_VAE(60, 30, 24, 1, 15, 256, 256, 256, 256, 0, 0, 0, False).cuda().train()
win = torch.randn(B, 45, 24, device="cuda")
for _ in range(5):
    m.loss_step(win, 1.0, beta=1.0, kloss=30, klmbda=0.1, bsize=B)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 30
for _ in range(N):
    m.loss_step(win, 1.0, beta=1.0, kloss=30, klmbda=0.1, bsize=B)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B}: host enqueue {1e3*(t1-t0)/N:.2f} ms/step, total {1e3*(t2-t0)/N:.2f} ms/step (fwd+loss+bwd only)")
