"""Training-set creation on the MI355X -- counterpart of vame/model/create_training.py.

`create_trainset(config)` turns the per-video pose series `<file>-PE-seq.npy` into `data/train/{train,test}_seq.npy` and the
per-video `<file>-PE-seq-clean.npy` that `train_model` / `pose_segmentation` read.  The reference does it with Python double
loops over every (frame, marker) entry (create_training.py:130-136, 225-236); here each O(N*F) pass is a float64 HIP kernel
(vame_amd/csrc/prep.hip) over a device-resident (F, N_total) buffer, and only scalars (per-file mean / std / IQR, the filter
weights, the 2 x L/2 edge columns of the Savitzky-Golay 'interp' mode) are computed on the host with the reference's own
numpy / scipy calls.  Output files are bit-identical to the reference's (tests/golden/prep_*.npz), including its two
peculiar NaN-fill rules:
  * aligned data: `interpol` on the whole file interpolates over the feature index, which makes every outlier of feature f
    the LAST valid sample of feature f (create_training.py:27-32, :145);
  * fixed (egocentric) data: outliers are interpolated across the features of the same frame (:236).
`check_parameter=True` plots instead of saving, like the reference (:198-199).
"""
import os
from pathlib import Path

import numpy as np
import torch

from .. import ops
from .. import _lib
from ..util.auxiliary import read_config


def _device():
    return _lib.device()          # looked up at call time (raises without an MI355X)


def _iqr(z):
    from scipy.stats import iqr
    return iqr(z)


def _resolve_empty_features(block, first_last):
    """Features without a single valid sample (never seen in practice): np.interp over the feature index puts them half way
    between the neighbouring populated features (clamped at the ends) -- done on the host for those rows only."""
    fl = first_last.cpu().numpy().reshape(-1, 2)
    empty = np.nonzero(np.isnan(fl[:, 1]))[0]
    if len(empty) == 0:
        return
    pop = np.nonzero(~np.isnan(fl[:, 1]))[0]
    if len(pop) == 0:
        raise ValueError("array of sample points is empty")          # what np.interp raises in the reference
    for f in empty:
        lo, hi = pop[pop < f], pop[pop > f]
        if len(lo) == 0:
            fill = fl[hi[0], 0]
        elif len(hi) == 0:
            fill = fl[lo[-1], 1]
        else:
            x0, x1, y0, y1 = float(lo[-1]), float(hi[0]), fl[lo[-1], 1], fl[hi[0], 0]
            fill = (y1 - y0) / (x1 - x0) * (f - x0) + y0
        block[f].fill_(fill)


def prepare_series(files_data, *, fixed, robust, iqr_factor, savgol_filter, savgol_length, savgol_order, device=None,
                   return_stages=False):
    """list of (F, N_i) float64 arrays -> (X_med (F', N_total) float64 host array, pos, info).

    Device pipeline: z-score + outlier mask -> NaN fill (per file) -> per-feature std / anchor removal (aligned) ->
    Savitzky-Golay along time.  `info` carries iqr values and the removed anchors."""
    dev = device or _device()
    F = files_data[0].shape[0]
    pos = np.concatenate([[0], np.cumsum([d.shape[1] for d in files_data])]).astype(np.int64)
    N = int(pos[-1])
    X = torch.empty(F, N, dtype=torch.float64, device=dev)
    first_last = torch.empty(F, 2, dtype=torch.float64, device=dev)
    n_empty = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = ops.prep_ws(F, dev)
    info = dict(iqr=[], anchors=None)
    for i, data in enumerate(files_data):
        if data.shape[0] != F:
            raise ValueError("all files must have the same number of features")
        Ni = data.shape[1]
        mean, sd = np.mean(data, axis=None), np.std(data, axis=None)
        cutoff = 0.0
        if robust:
            # the IQR is a property of the z-scored file; (x - mean) / sd is monotone, so it follows from the raw percentiles
            # only approximately in floating point -- compute it from the z-scores like the reference (host, O(n) partition)
            iqr_val = _iqr((data.T - mean) / sd)
            info['iqr'].append(iqr_val)
            cutoff = iqr_factor * iqr_val
            print("IQR value: %.2f, IQR cutoff: %.2f" % (iqr_val, cutoff))
        raw = torch.from_numpy(np.ascontiguousarray(data, dtype=np.float64)).to(dev)
        ops.prep_zscore_mask(raw, F, Ni, Ni, mean, sd, cutoff, robust, X, N, z_off=int(pos[i]))
        if robust:
            if fixed:
                ops.prep_fill_across_features(X, F, Ni, N, n_empty, z_off=int(pos[i]))
            else:
                ops.prep_fill_last_valid(X, F, Ni, N, first_last, ws, z_off=int(pos[i]))
                _resolve_empty_features(X[:, int(pos[i]):int(pos[i + 1])], first_last)
    if fixed and robust and int(n_empty.item()) > 0:
        raise ValueError("array of sample points is empty")          # np.interp on an all-outlier frame (create_training.py:236)
    stages = dict(filled=X.cpu().numpy()) if return_stages else None
    keep = list(range(F))
    if not fixed:
        mean_f = torch.empty(F, dtype=torch.float64, device=dev)
        std_f = torch.empty(F, dtype=torch.float64, device=dev)
        ops.prep_rowstats(X, F, N, N, mean_f, std_f, ws)
        d = std_f.cpu().numpy()
        s = np.sort(d)
        if s[0] == s[1]:
            a = np.where(d == s[0])[0]
            a1, a2 = int(a[0]), int(a[1])
        else:
            a1, a2 = int(np.where(d == s[0])[0][0]), int(np.where(d == s[1])[0][0])
        anchor_1, anchor_2 = (a1, a2) if a1 > a2 else (a2, a1)          # create_training.py:166-172: delete the larger index first
        del keep[anchor_1]
        del keep[anchor_2]
        info['anchors'] = (anchor_1, anchor_2)
        X = X[keep].contiguous()
    Fk = X.shape[0]
    if savgol_filter:
        from scipy.signal import savgol_coeffs, savgol_filter as sg
        w = np.ascontiguousarray(savgol_coeffs(savgol_length, savgol_order)[::-1])   # convolve1d reverses the weights
        half = savgol_length // 2
        if not np.all(np.abs(w - w[::-1]) <= np.finfo(np.float64).eps):
            raise NotImplementedError("asymmetric Savitzky-Golay weights")       # never for deriv = 0
        Y = torch.empty_like(X)
        ops.prep_savgol(X, Fk, N, N, torch.from_numpy(w).to(dev), savgol_length, Y, N)
        if half > 0:
            # mode='interp': the first / last L//2 samples come from a polynomial fitted to the first / last L samples
            span = min(N, 2 * savgol_length)
            left = sg(X[:, :span].cpu().numpy(), savgol_length, savgol_order)[:, :half]
            right = sg(X[:, N - span:].cpu().numpy(), savgol_length, savgol_order)[:, -half:]
            Y[:, :half] = torch.from_numpy(np.ascontiguousarray(left)).to(dev)
            Y[:, N - half:] = torch.from_numpy(np.ascontiguousarray(right)).to(dev)
        X = Y
    out = X.cpu().numpy()
    if return_stages:
        info['stages'] = stages
    return out, pos, info


def _load_files(cfg, files):
    return [np.load(os.path.join(cfg['project_path'], "data", f, f + '-PE-seq.npy')) for f in files]


def _save_split(cfg, files, X_med, pos, testfraction):
    num_frames = X_med.shape[1]
    test = int(num_frames * testfraction)
    z_test, z_train = X_med[:, :test], X_med[:, test:]
    np.save(os.path.join(cfg['project_path'], "data", "train", 'train_seq.npy'), z_train)
    np.save(os.path.join(cfg['project_path'], "data", "train", 'test_seq.npy'), z_test)
    for i, file in enumerate(files):
        np.save(os.path.join(cfg['project_path'], "data", file, file + '-PE-seq-clean.npy'), X_med[:, pos[i]:pos[i + 1]])
    print('Lenght of train data: %d' % z_train.shape[1])
    print('Lenght of test data: %d' % z_test.shape[1])


def _plot_check(cfg, files_data, X_med, iqr_val):
    """Overview plots of plot_check_parameter (create_training.py:34-95): z-scored original vs filtered signal + IQR cutoff."""
    from .evaluate import _pyplot
    plt = _pyplot()
    data = files_data[0]
    orig = ((data.T - np.mean(data, axis=None)) / np.std(data, axis=None)).T
    cut = cfg['iqr_factor'] * iqr_val
    n = X_med.shape[1]
    lo = np.random.choice(n) if n > 1000 else 0
    sl = slice(lo, lo + 1000) if n > 1000 else slice(None)
    figs = []
    for arr, title in ((orig, "Full Signal z-scored"), (X_med[:, sl], "Filtered signal z-scored"), (orig[:, sl], "Original signal z-scored")):
        fig = plt.figure()
        plt.plot(arr.T)
        plt.axhline(y=cut, color='r', linestyle='--', label="IQR cutoff")
        plt.axhline(y=-cut, color='r', linestyle='--')
        plt.title(title)
        plt.legend()
        figs.append(fig)
    print("Please run the function with check_parameter=False if you are happy with the results")
    return figs


def _traindata(cfg, files, testfraction, savgol_filter, check_parameter, fixed):
    if check_parameter:
        files = [files[0]]
    for file in files:
        print("z-scoring of file %s" % file)
    datas = _load_files(cfg, files)
    X_med, pos, info = prepare_series(datas, fixed=fixed, robust=cfg['robust'] == True, iqr_factor=cfg['iqr_factor'],   # noqa: E712
                                      savgol_filter=savgol_filter, savgol_length=cfg['savgol_length'],
                                      savgol_order=cfg['savgol_order'])
    if check_parameter:
        _plot_check(cfg, datas, X_med, info['iqr'][0])      # like the reference this needs cfg['robust'] (iqr_val otherwise unbound)
    else:
        _save_split(cfg, files, X_med, pos, testfraction)
    return X_med


def traindata_aligned(cfg, files, testfraction, num_features, savgol_filter, check_parameter):
    return _traindata(cfg, files, testfraction, savgol_filter, check_parameter, fixed=False)


def traindata_fixed(cfg, files, testfraction, num_features, savgol_filter, check_parameter):
    return _traindata(cfg, files, testfraction, savgol_filter, check_parameter, fixed=True)


def create_trainset(config, check_parameter=False):
    cfg = read_config(Path(config).resolve())
    fixed = cfg['egocentric_data']
    os.makedirs(os.path.join(cfg['project_path'], 'data', 'train', ""), exist_ok=True)
    files = []
    if cfg['all_data'] == 'No':
        for file in cfg['video_sets']:
            if input("Do you want to train on " + file + "? yes/no: ") == 'yes':
                files.append(file)
    else:
        files = list(cfg['video_sets'])
    print("Creating training dataset...")
    if cfg['robust'] == True:   # noqa: E712
        print("Using robust setting to eliminate outliers! IQR factor: %d" % cfg['iqr_factor'])
    if fixed == False:          # noqa: E712
        print("Creating trainset from the vame.egocentrical_alignment() output ")
        traindata_aligned(cfg, files, cfg['test_fraction'], cfg['num_features'], cfg['savgol_filter'], check_parameter)
    else:
        print("Creating trainset from the vame.csv_to_numpy() output ")
        traindata_fixed(cfg, files, cfg['test_fraction'], cfg['num_features'], cfg['savgol_filter'], check_parameter)
    if check_parameter == False:  # noqa: E712
        print("A training and test set has been created. Next step: vame.train_model()")
