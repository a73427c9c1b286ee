"""Sliding-window batcher.

Reference: vame/model/dataloader.py:18-56 (`SEQUENCE_DATASET`) + torch DataLoader collate.
`SEQUENCE_DATASET` keeps the reference's constructor, `__len__` and `__getitem__` semantics
(random start, index ignored, float64 (F, 2T) item) for API users.  Training uses
`DeviceWindowLoader`: the z-scored series is resident in HBM as fp32 and a HIP gather kernel
cuts B windows per step (out[b,t,f] = Xn[f, start_b+t]) -- bit-identical to the reference's
per-window (x-mean)/std in float64 followed by the cast to float32 at rnn_vae.py:111-115.
"""
import os

import numpy as np
import torch
from torch.utils.data.dataset import Dataset

from .. import ops


def _series_statistics(folder, series, compute):
    """Global scalar mean / std of the TRAIN series, kept next to it as seq_mean.npy / seq_std.npy (the on-disk contract of
    dataloader.py:27-35): computed and written by the first training run, read by every later run and by the test split."""
    files = [folder + name for name in ("seq_mean.npy", "seq_std.npy")]
    if compute and not os.path.exists(os.path.join(folder, "seq_mean.npy")):
        print("Compute mean and std for temporal dataset.")
        stats = (np.mean(series), np.std(series))
        for f, v in zip(files, stats):
            np.save(f, v)
        return stats
    return tuple(np.load(f) for f in files)


class SEQUENCE_DATASET(Dataset):
    """Same constructor, attributes (`X` (F, N), `mean`, `std`, `data_points`, `temporal_window`) and item semantics as the reference's class."""

    def __init__(self, path_to_file, data, train, temporal_window):
        series = np.load(path_to_file + data)
        self.X = series.T if series.shape[0] > series.shape[1] else series        # stored feature-major whatever the file's orientation (:22-23)
        self.temporal_window = temporal_window
        self.data_points = self.X.shape[1]
        self.mean, self.std = _series_statistics(path_to_file, self.X, train)
        print('Initialize %s data. Datapoints %d' % ('train' if train else 'test', self.data_points))

    def __len__(self):
        return self.data_points

    def __getitem__(self, index):
        start = np.random.choice(self.data_points - self.temporal_window)
        sequence = self.X[:, start:start + self.temporal_window]
        return torch.from_numpy((sequence - self.mean) / self.std)

    def normalised_f32(self):
        """(F,N) float32 of the whole z-scored series (normalised in float64 first, like __getitem__)."""
        return ((self.X - self.mean) / self.std).astype(np.float32)


class DeviceWindowLoader:
    """Iterable over floor(N / batch_size) batches of (B, L, F) fp32 device windows, L = window length kept.

    Window starts come from the global numpy RNG, `np.random.randint(0, N - 2T, size=B)`, which is
    stream-equivalent to the B scalar `np.random.choice(N - 2T)` calls the reference's DataLoader
    makes per batch (dataloader.py:49; tests/golden/batcher.npz).  With several ranks each rank keeps
    its own slice [rank*B, (rank+1)*B) of a B*world draw (disjoint slices of one stream when the ranks
    seed numpy identically, independent draws otherwise -- the reference never seeds numpy).
    """

    def __init__(self, dataset: SEQUENCE_DATASET, batch_size, keep, device, rank=0, world=1):
        self.ds, self.B, self.L, self.dev = dataset, int(batch_size), int(keep), device
        self.N, self.F = dataset.data_points, dataset.X.shape[0]
        self.T2 = dataset.temporal_window
        self.rank, self.world = rank, world
        self.Xn = torch.from_numpy(dataset.normalised_f32()).to(device).contiguous()
        self.n_batches = self.N // (self.B * world)
        # window starts travel through two pinned staging buffers so the 8*B-byte upload never blocks the host
        pin = device.type == "cuda"
        self._stage = [torch.empty(self.B, dtype=torch.int64, pin_memory=pin) for _ in range(2)]
        self._ev = [None, None]
        self._k = 0

    def __len__(self):
        return self.n_batches

    def draw_starts(self):
        s = np.random.randint(0, self.N - self.T2, size=self.B * self.world)
        return s[self.rank * self.B:(self.rank + 1) * self.B]

    def gather(self, starts):
        k = self._k = self._k ^ 1
        if self._ev[k] is not None:
            self._ev[k].synchronize()                   # the copy that last used this staging buffer has finished
        self._stage[k].copy_(torch.from_numpy(np.ascontiguousarray(starts, dtype=np.int64)))
        st = self._stage[k].to(self.dev, non_blocking=True)
        if self.dev.type == "cuda":
            self._ev[k] = torch.cuda.Event()
            self._ev[k].record()
        out = torch.empty(self.B, self.L, self.F, device=self.dev)
        ops.window_gather(self.Xn, self.N, self.F, st, 0, self.B, self.L, out)
        return out

    # ---- the same batcher in two halves for a captured step graph (rnn_vae.GraphedTrainStep): the window starts travel to a FIXED device
    # buffer outside the graph (the only host -> device traffic of a step), the gather kernel inside it reads that buffer and writes a
    # fixed window buffer
    def upload_starts(self, starts):
        if getattr(self, "_starts_dev", None) is None:
            self._starts_dev = torch.empty(self.B, dtype=torch.int64, device=self.dev)
        k = self._k = self._k ^ 1
        if self._ev[k] is not None:
            self._ev[k].synchronize()
        self._stage[k].copy_(torch.from_numpy(np.ascontiguousarray(starts, dtype=np.int64)))
        self._starts_dev.copy_(self._stage[k], non_blocking=True)
        if self.dev.type == "cuda":
            self._ev[k] = torch.cuda.Event()
            self._ev[k].record()

    def gather_static(self):
        if getattr(self, "_starts_dev", None) is None:
            self._starts_dev = torch.empty(self.B, dtype=torch.int64, device=self.dev)
        if getattr(self, "_win_static", None) is None:
            self._win_static = torch.empty(self.B, self.L, self.F, device=self.dev)
        ops.window_gather(self.Xn, self.N, self.F, self._starts_dev, 0, self.B, self.L, self._win_static)
        return self._win_static

    def __iter__(self):
        for _ in range(self.n_batches):
            yield self.gather(self.draw_starts())
