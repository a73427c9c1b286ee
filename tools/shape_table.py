#!/usr/bin/env python3
"""Train-step throughput over batch sizes (H = 256) and hidden sizes (batch 4096): the "no cliff" table of profiles/r05_shape_table.txt.
   python tools/shape_table.py [batch | hidden | both]
Each row: windows/s of gather + loss_step + Adam (HIP events over `steps` steps after a warm-up; a replayed hipGraph up to batch 1024 like
train()), which GRU kernels the engine chose, and the row's ratio to the better of its neighbours (a cliff shows as a ratio well below 1)."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch  # noqa: E402

import bench  # noqa: E402
from vame_amd import ops  # noqa: E402
from vame_amd.model.dataloader import DeviceWindowLoader  # noqa: E402
from vame_amd.model.rnn_model import RNN_VAE  # noqa: E402
from vame_amd.model.rnn_vae import FusedAdamAMSGrad, GraphedTrainStep  # noqa: E402

dev = torch.device("cuda")
what = sys.argv[1] if len(sys.argv) > 1 else "both"


def run(H, T, B):
    torch.manual_seed(19)
    model = RNN_VAE(2 * T, bench.Z, bench.F, 1, bench.FS, H, H, H, H, 0, 0, 0, False).to(dev).train()
    if os.environ.get("SHAPE_ENGINE"):           # e.g. SHAPE_ENGINE=coop_rounds:4,wide:0  (A/B of engine options)
        model.engine_options = {k: int(v) for k, v in (kv.split(":") for kv in os.environ["SHAPE_ENGINE"].split(","))}
    opt = FusedAdamAMSGrad(model, lr=5e-4)
    loader = DeviceWindowLoader(bench._SynthDataset(T), B, T + bench.FS, dev, rank=0, world=1)
    kw = dict(kl_weight=1.0, beta=1.0, kloss=bench.Z, klmbda=0.1, bsize=B)
    calls = set()
    saved = {}
    for name in ("gru_seq_fwd", "gru_coop_fwd", "gru_wide_fwd"):
        saved[name] = getattr(ops, name)
        setattr(ops, name, (lambda n, f: (lambda *a, **k: (calls.add(n), f(*a, **k))[1]))(name, saved[name]))
    if B <= 1024:
        g = GraphedTrainStep(model, opt, loader, torch.zeros(6, dtype=torch.float64, device=dev), **kw)
        step = lambda: g(loader.draw_starts())  # noqa: E731
    else:
        def step():
            model.loss_step(loader.gather(loader.draw_starts()), **kw)
            opt.step()
    steps = 30 if B <= 1024 else (12 if H <= 256 else 6)
    for _ in range(5):
        step()
    for name, f in saved.items():
        setattr(ops, name, f)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    eng = model._engine
    eng.check_async_errors()
    kern = "+".join(sorted(c.replace("gru_", "").replace("_fwd", "") for c in calls)) or ("stepwise" if eng.stepwise else "?")
    del model, opt, loader
    bench.release_leg(True)
    return B / ms * 1e3, ms, kern


def table(rows, label):
    res = [(k,) + run(*cfg) for k, cfg in rows]
    print(f"--- {label}")
    for i, (k, wps, ms, kern) in enumerate(res):
        nb = [res[j][1] for j in (i - 1, i + 1) if 0 <= j < len(res)]
        print(f"{k:>8}  {wps:10.0f} windows/s  {ms:8.3f} ms/step  GRU kernels: {kern:18s}")
    return res


if what in ("batch", "both"):
    bs = tuple(int(b) for b in os.environ["SHAPE_BATCHES"].split(",")) if os.environ.get("SHAPE_BATCHES") else (128, 256, 384, 512, 768, 1024, 1280, 1536, 2048, 3072, 4096)
    res = table([(b, (256, 30, b)) for b in bs], "batch sweep, H = 256, T = 30 (monotone = no cliff)")
    bad = [res[i][0] for i in range(1, len(res)) if res[i][1] < res[i - 1][1]]
    print("not monotone at:", bad if bad else "none")
if what in ("hidden", "both"):
    hs = (256, 288, 320, 352, 384, 416, 448, 480, 512)
    res = table([(h, (h, 30, 4096)) for h in hs], "hidden-size sweep, batch 4096, T = 30")
    # compare in flop terms: windows/s x MFLOP per window
    tf = [(h, w * bench.train_mflop_per_window(h, 30) / 1e6) for h, w, _, _ in res]
    for i, (h, t) in enumerate(tf):
        nb = max(tf[j][1] for j in (i - 1, i + 1) if 0 <= j < len(tf))
        print(f"H={h}: {t:6.1f} TF  vs better neighbour {t / nb:5.2f}")
