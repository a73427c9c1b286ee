#!/usr/bin/env python3
"""k-means over synthetic latent vectors: MI355X (KMeansHIP) vs scikit-learn on the host."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vame_amd.analysis.kmeans_hip import KMeansHIP
from vame_amd import ops
N, D, K = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 30, 15
rng = np.random.default_rng(0)
cent = rng.standard_normal((K, D)) * 2
X = (cent[rng.integers(0, K, N)] + rng.standard_normal((N, D))).astype(np.float32)
Xd = torch.from_numpy(X).cuda()
km = KMeansHIP(K, n_init=2, random_state=42)
km.fit(Xd[:100000])                                   # warm-up
torch.cuda.synchronize(); t0 = time.perf_counter()
km = KMeansHIP(K, n_init=20, random_state=42).fit(Xd)
torch.cuda.synchronize(); t_gpu = time.perf_counter() - t0
# E-step kernel alone: HBM roofline (4*D bytes in + 4+4 bytes out per row, no one-hot)
C = torch.from_numpy(km.cluster_centers_).cuda()
lab = torch.empty(N, dtype=torch.int32, device="cuda"); d2 = torch.empty(N, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3): ops.kmeans_assign(Xd, N, D, C, K, lab, d2)
e0.record()
for _ in range(20): ops.kmeans_assign(Xd, N, D, C, K, lab, d2)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
out = dict(N=N, D=D, K=K, n_init=20, gpu_fit_seconds=round(t_gpu, 3), gpu_inertia=km.inertia_, gpu_iters_best=km.n_iter_,
           assign_kernel_ms=round(ms, 4), assign_GBps=round(N * (4 * D + 8) / ms / 1e6, 1), assign_frac_of_8TBps=round(N * (4 * D + 8) / ms / 1e6 / 8000, 3))
if "--sklearn" in sys.argv:
    from sklearn.cluster import KMeans
    t0 = time.perf_counter()
    sk = KMeans(init="k-means++", n_clusters=K, random_state=42, n_init=20).fit(X)
    out.update(sklearn_fit_seconds=round(time.perf_counter() - t0, 2), sklearn_inertia=float(sk.inertia_), threads=len(os.sched_getaffinity(0)))
print(json.dumps(out))
