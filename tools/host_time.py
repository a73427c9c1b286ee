"""Host enqueue time vs wall time of the whole train step (gather + forward + loss + backward + Adam) at a given batch size, and the same
with the host deliberately throttled out of the picture: N steps enqueued while the GPU is held by a long sleep kernel show the pure GPU
time of a step.  At the reference's stock batch (256) the step is HOST-bound.  usage: python tools/host_time.py [batch ...]"""
import sys, time, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import numpy as np, torch
import bench
from vame_amd.model.dataloader import DeviceWindowLoader
from vame_amd.model.rnn_model import RNN_VAE
from vame_amd.model.rnn_vae import FusedAdamAMSGrad
dev = torch.device("cuda")
for B in [int(a) for a in sys.argv[1:]] or [256]:
    torch.manual_seed(19)
    m = RNN_VAE(60, 30, 24, 1, 15, 256, 256, 256, 256, 0, 0, 0, False).cuda().train()
    opt = FusedAdamAMSGrad(m, lr=5e-4)
    loader = DeviceWindowLoader(bench._SynthDataset(30), B, 45, dev, rank=0, world=1)
    def step():
        win = loader.gather(loader.draw_starts()); m.loss_step(win, 1.0, beta=1.0, kloss=30, klmbda=0.1, bsize=B); opt.step()
    for _ in range(10): step()
    torch.cuda.synchronize()
    N = 30
    t0 = time.perf_counter()
    for _ in range(N): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    # GPU time alone: hold the stream with a ~0.15 s sleep kernel, enqueue N steps behind it, time them with events
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(0.15 * 2.4e9))
    e0.record()
    for _ in range(N): step()
    e1.record(); torch.cuda.synchronize()
    print(f"B={B}: host enqueue {1e3 * (t1 - t0) / N:.3f} ms/step, wall {1e3 * (t2 - t0) / N:.3f} ms/step, GPU alone (queue pre-filled) {e0.elapsed_time(e1) / N:.3f} ms/step", flush=True)
    del m, opt, loader
    bench.release_leg(True)
