"""CPU restatement (numpy, float64) of the arithmetic of vame/model/create_training.py -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product path
(vame_amd/model/create_training.py) runs the HIP kernels in vame_amd/csrc/prep.hip and never calls in here.

Pinned against the reference: tests/golden/make_golden.py runs the reference's traindata_aligned / traindata_fixed
(imported by path) on small synthetic projects and stores inputs + outputs in tests/golden/prep_*.npz;
tests/test_oracle.py checks this file against them bit for bit.

What the reference computes per file (create_training.py:107-147 aligned, :210-239 fixed):
  z = (data.T - mean(data)) / std(data)                       global scalar mean / std of the file, float64
  robust: iqr_val = scipy.stats.iqr(z); entries with z > f*iqr or z < -f*iqr become NaN, then `interpol`:
    * aligned (:145): interpol on the whole (N, F) array.  interpol transposes to (F, N), takes
      x = nonzero(mask)[0] -- the FEATURE index of every entry, not its frame -- and calls
      np.interp(x_nan, x_valid, values_valid).  With x_valid non-decreasing and full of repeats, np.interp returns, for a
      NaN of feature f, the LAST valid sample (in time) of feature f  (binary search lands on the last j with xp[j] <= f and
      x == xp[j]).  If feature f has no valid sample at all it interpolates half way between the last valid sample of the
      previous populated feature and the first valid sample of the next one (clamped at the ends).
    * fixed (:236): interpol on each frame's 1-D feature vector: linear interpolation ACROSS FEATURES inside the frame,
      clamped to the first / last valid feature value at the ends.
  then files are concatenated in time; aligned only: the two features with the smallest std over time are deleted (:153-176).
  savgol: scipy.signal.savgol_filter(X (F,N), savgol_length, savgol_order) along time, mode='interp' (:180-183, :244-247).
  split: test = first int(N*test_fraction) frames, train = the rest (:185-189).
"""
import numpy as np


def zscore(data):
    """(F, N) file array -> (N, F) z-scored like create_training.py:112-114."""
    return (data.T - np.mean(data, axis=None)) / np.std(data, axis=None)


def iqr_value(z):
    """scipy.stats.iqr default = np.percentile(z, 75) - np.percentile(z, 25) with linear interpolation."""
    q = np.percentile(z, [25, 75])
    return q[1] - q[0]


def mask_outliers(z, cutoff):
    out = z.copy()
    out[(z > cutoff) | (z < -cutoff)] = np.nan
    return out


def fill_aligned(z_nf):
    """interpol() on the whole (N, F) array, closed form of the np.interp call described above."""
    y = z_nf.T.copy()                                    # (F, N)
    F = y.shape[0]
    valid = ~np.isnan(y)
    has = valid.any(axis=1)
    if not valid.all():
        last = np.full(F, np.nan)
        first = np.full(F, np.nan)
        for f in range(F):
            if has[f]:
                idx = np.nonzero(valid[f])[0]
                last[f], first[f] = y[f, idx[-1]], y[f, idx[0]]
        pop = np.nonzero(has)[0]
        for f in range(F):
            if valid[f].all():
                continue
            if has[f]:
                fill = last[f]
            else:                                        # feature without any valid sample
                lo, hi = pop[pop < f], pop[pop > f]
                if len(lo) == 0:
                    fill = first[hi[0]]                  # np.interp clamps to fp[0]
                elif len(hi) == 0:
                    fill = last[lo[-1]]                  # ... and to fp[-1]
                else:
                    x0, x1, y0, y1 = float(lo[-1]), float(hi[0]), last[lo[-1]], first[hi[0]]
                    fill = (y1 - y0) / (x1 - x0) * (f - x0) + y0
            y[f, ~valid[f]] = fill
    return y.T


def fill_fixed(z_nf):
    """interpol() on every frame's feature vector: np.interp across the feature index (create_training.py:236)."""
    out = z_nf.copy()
    F = out.shape[1]
    xs = np.arange(F, dtype=np.float64)
    for i in np.nonzero(np.isnan(out).any(axis=1))[0]:
        row = out[i]
        nan = np.isnan(row)
        row[nan] = np.interp(xs[nan], xs[~nan], row[~nan])
    return out


def savgol_coeffs(length, order):
    """Least-squares smoothing weights for the window centre (scipy.signal.savgol_coeffs, deriv=0), via the same lstsq."""
    half = length // 2
    x = np.arange(-half, length - half, dtype=float)[::-1]
    A = x ** np.arange(order + 1).reshape(-1, 1)
    yv = np.zeros(order + 1)
    yv[0] = 1.0
    coeffs, *_ = np.linalg.lstsq(A, yv, rcond=None)
    return coeffs


def anchors_to_delete(X_nf):
    """create_training.py:153-176 -> (anchor_1, anchor_2) with anchor_1 > anchor_2, deleted in that order."""
    d = np.std(X_nf.T, axis=1)
    s = np.sort(d)
    if s[0] == s[1]:
        a = np.where(d == s[0])[0]
        a1, a2 = int(a[0]), int(a[1])
    else:
        a1, a2 = int(np.where(d == s[0])[0][0]), int(np.where(d == s[1])[0][0])
    return (a1, a2) if a1 > a2 else (a2, a1)


def traindata(files_data, *, fixed, robust, iqr_factor, savgol_filter, savgol_length, savgol_order, test_fraction):
    """files_data: list of (F, N_i) arrays -> dict(train, test, clean=[per-file (F', N_i)], pos)."""
    import scipy.signal
    parts, pos = [], [0]
    for data in files_data:
        z = zscore(data)
        if robust:
            z = mask_outliers(z, iqr_factor * iqr_value(z))
            z = fill_fixed(z) if fixed else fill_aligned(z)
        parts.append(z)
        pos.append(pos[-1] + data.shape[1])
    X = np.concatenate(parts, axis=0)
    if not fixed:
        a1, a2 = anchors_to_delete(X)
        X = np.delete(np.delete(X, a1, 1), a2, 1)
    X = X.T
    X_med = scipy.signal.savgol_filter(X, savgol_length, savgol_order) if savgol_filter else X
    test = int(X_med.shape[1] * test_fraction)
    return dict(train=X_med[:, test:], test=X_med[:, :test], clean=[X_med[:, pos[i]:pos[i + 1]] for i in range(len(files_data))], pos=pos)


def traindata_as_written_seconds(data, *, fixed, iqr_factor=4, savgol_length=5, savgol_order=2):
    """Wall time of the reference's per-file work in the form it is written in (create_training.py:107-147 / 210-239: a
    Python loop over every frame and marker for the outlier test, np.interp fills, scipy savgol) -- the CPU baseline of
    tools/prep_bench.py.  Returns (seconds, result)."""
    import time
    import scipy.signal
    t0 = time.perf_counter()
    z = zscore(data)
    cutoff = iqr_factor * iqr_value(z)
    for i in range(z.shape[0]):
        for marker in range(z.shape[1]):
            if z[i, marker] > cutoff:
                z[i, marker] = np.nan
            elif z[i, marker] < -cutoff:
                z[i, marker] = np.nan
        if fixed:
            row = z[i, :]
            nan = np.isnan(row)
            if nan.any():
                idx = np.arange(row.shape[0])
                row[nan] = np.interp(idx[nan], idx[~nan], row[~nan])
    if not fixed:
        y = z.T
        nans = np.isnan(y)
        y[nans] = np.interp(nans.nonzero()[0], (~nans).nonzero()[0], y[~nans])
        z = y.T
    out = scipy.signal.savgol_filter(z.T, savgol_length, savgol_order)
    return time.perf_counter() - t0, out
