#!/usr/bin/env python3
"""create_trainset preparation on the MI355X vs the reference's CPU form.

usage: python tools/prep_bench.py [frames_per_file] [--cpu-frames N]
Times vame_amd.model.create_training.prepare_series on two synthetic pose files (26 features, aligned rule + Savitzky-Golay),
checks it against the numpy oracle on the same input, reports per-kernel HBM rates (HIP events) and the reference-form CPU time
on a bounded sample.  One JSON line on stdout."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import prep_oracle as po
from vame_amd import ops
from vame_amd.model.create_training import prepare_series

N = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 1_000_000
cpu_frames = int(sys.argv[sys.argv.index("--cpu-frames") + 1]) if "--cpu-frames" in sys.argv else 40_000
F = 26
dev = torch.device("cuda")


def synth(seed, n):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((F, n)).cumsum(axis=1) * 0.05 + rng.standard_normal((F, n))
    idx = rng.integers(0, n, size=n // 50)
    X[rng.integers(0, F, size=len(idx)), idx] *= 40.0
    X[3] = 0.0
    X[7] = 0.0
    return X


datas = [synth(1, N), synth(2, N)]
kw = dict(fixed=False, robust=True, iqr_factor=4, savgol_filter=True, savgol_length=5, savgol_order=2)
prepare_series([d[:, :5000] for d in datas], device=dev, **kw)          # warm-up (library load, allocator)
torch.cuda.synchronize()
t0 = time.perf_counter()
out, pos, info = prepare_series(datas, device=dev, **kw)
torch.cuda.synchronize()
t_gpu = time.perf_counter() - t0
ref = po.traindata(datas, test_fraction=0.1, **kw)
same = bool(np.array_equal(out, np.concatenate([ref["test"], ref["train"]], axis=1)))


def rate(fn, nbytes, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return dict(us=round(ms * 1e3, 1), GBps=round(nbytes / ms / 1e6, 1), frac_of_8TBps=round(nbytes / ms / 1e6 / 8000, 3))


Nt = 2 * N
x = torch.from_numpy(datas[0]).to(dev)
z = torch.empty(F, Nt, dtype=torch.float64, device=dev)
y = torch.empty_like(z)
fl = torch.empty(F, 2, dtype=torch.float64, device=dev)
m, s = torch.empty(F, dtype=torch.float64, device=dev), torch.empty(F, dtype=torch.float64, device=dev)
w = torch.from_numpy(np.ascontiguousarray(po.savgol_coeffs(5, 2)[::-1])).to(dev)
ws = ops.prep_ws(F, dev)
kern = dict(
    zscore_mask=rate(lambda: ops.prep_zscore_mask(x, F, N, N, 0.1, 1.3, 5.0, True, z, Nt), 16 * F * N),
    fill_last_valid=rate(lambda: ops.prep_fill_last_valid(z, F, N, Nt, fl, ws), 16 * F * N),
    rowstats=rate(lambda: ops.prep_rowstats(z, F, Nt, Nt, m, s, ws), 16 * F * Nt),
    savgol=rate(lambda: ops.prep_savgol(z, F, Nt, Nt, w, 5, y, Nt), 16 * F * Nt),
)
t_cpu, _ = po.traindata_as_written_seconds(datas[0][:, :cpu_frames].copy(), fixed=False)
print(json.dumps(dict(
    metric="create_trainset preparation, frames/s (26 features, aligned rule, robust, savgol 5/2)", frames=Nt,
    gpu_seconds_end_to_end=round(t_gpu, 3), gpu_frames_per_s=round(Nt / t_gpu, 1), bit_identical_to_oracle=same,
    note="end to end = host mean/std/iqr + H2D + 5 kernel passes + D2H of the result; kernels are HBM-bound float64 passes",
    kernels=kern,
    cpu_baseline=dict(kind="port", form="reference loop form (create_training.py:130-147), numpy/scipy, 1 core", sample_frames=cpu_frames,
                      seconds=round(t_cpu, 2), frames_per_s=round(cpu_frames / t_cpu, 1)),
    speedup=round((Nt / t_gpu) / (cpu_frames / t_cpu), 1))))
