"""Test-set evaluation on the MI355X path -- counterpart of vame/model/evaluate.py.

The reference's `evaluate_model` is a thin plotting layer over one eval-mode forward pass of the trained
RNN-VAE on a batch of 64 random test windows (`plot_reconstruction`, evaluate.py:29-81) and over the eight
loss arrays `train_model` wrote (`plot_loss`, :84-115).  Here the forward pass runs through the HIP kernels
(`RNN_VAE.forward` in eval mode: mu feeds the decoders, rnn_model.py:75-76) and the test windows are cut on
the device by the gather kernel; the numbers that get plotted are returned by `reconstruct_test_batch` so
they can be tested without looking at a PNG.

Kept from the reference: file names under `<project>/model/evaluate/`, TEST_BATCH_SIZE = 64 (:130), five
panels, the `suffix` naming for snapshots, the window draw (`np.random` global stream, one start per sample).
Deliberate difference: with `use_snapshots=True` the reference's GPU branch reloads the *best* model for
every snapshot (:150 ignores `snapshot`); here the snapshot file itself is loaded, which is what :155-156
(the CPU branch) intends.
"""
import os
from pathlib import Path

import numpy as np
import torch

from .. import _lib
from ..util.auxiliary import read_config
from .dataloader import SEQUENCE_DATASET, DeviceWindowLoader
from .rnn_model import RNN_VAE

TEST_BATCH_SIZE = 64
_LOSS_CURVES = (('train_losses_', 'Train-Loss'), ('test_losses_', 'Test-Loss'), ('mse_train_losses_', 'MSE-Train-Loss'),
                ('mse_test_losses_', 'MSE-Test-Loss'), ('kmeans_losses_', 'KMeans-Loss'), ('kl_losses_', 'KL-Loss'),
                ('fut_losses_', 'Prediction-Loss'))


def _device():
    return _lib.device()          # looked up at call time (raises without an MI355X)


def _pyplot():
    import matplotlib
    if not os.environ.get("DISPLAY"):
        matplotlib.use("Agg", force=False)
    from matplotlib import pyplot as plt
    return plt


def reconstruct_test_batch(model, windows, seq_len_half, FUTURE_DECODER, FUTURE_STEPS):
    """Eval-mode forward of `windows` (B, >= T+FS, F) device fp32 -> dict of host arrays.

    Keys: data (B,T,F), data_tilde (B,T,F), latent/mu/logvar (B,Z) and, with a future decoder,
    fut_orig / fut (B,FS,F)  (evaluate.py:35-56).
    """
    data = windows[:, :seq_len_half, :].contiguous()
    out = {'data': data.cpu().numpy()}
    with torch.no_grad():
        res = model(data)
    if FUTURE_DECODER:
        x_tilde, future, latent, mu, logvar = res
        out['fut_orig'] = windows[:, seq_len_half:seq_len_half + FUTURE_STEPS, :].cpu().numpy()
        out['fut'] = future.detach().cpu().numpy()
    else:
        x_tilde, latent, mu, logvar = res
    out['data_tilde'] = x_tilde.detach().cpu().numpy()
    out['latent'], out['mu'], out['logvar'] = (t.detach().cpu().numpy() for t in (latent, mu, logvar))
    return out


def plot_reconstruction(filepath, test_loader, seq_len_half, model, model_name, FUTURE_DECODER, FUTURE_STEPS, suffix=None):
    """`test_loader`: a DeviceWindowLoader (device windows) or any iterable of (B, F, 2T) batches like the
    reference's DataLoader over SEQUENCE_DATASET."""
    x = next(iter(test_loader))
    if not isinstance(test_loader, DeviceWindowLoader):
        dev = next(model.parameters()).device
        x = x.permute(0, 2, 1).to(dtype=torch.float32).to(dev)
    r = reconstruct_test_batch(model, x, seq_len_half, FUTURE_DECODER, FUTURE_STEPS)
    plt = _pyplot()
    n_panels = min(5, r['data'].shape[0])
    if FUTURE_DECODER:
        fig, axs = plt.subplots(2, 5)
        fig.suptitle('Reconstruction [top] and future prediction [bottom] of input sequence')
        for i in range(n_panels):
            axs[0, i].plot(r['data'][i], color='k', label='Sequence Data')
            axs[0, i].plot(r['data_tilde'][i], color='r', linestyle='dashed', label='Sequence Reconstruction')
            axs[1, i].plot(r['fut_orig'][i], color='k')
            axs[1, i].plot(r['fut'][i], color='r', linestyle='dashed')
        axs[0, 0].set(xlabel='time steps', ylabel='reconstruction')
        axs[1, 0].set(xlabel='time steps', ylabel='predction')
        target = os.path.join(filepath, 'evaluate', 'Future_Reconstruction.png')
        fig.savefig(target)
    else:
        fig, axs = plt.subplots(1, 5)
        fig.suptitle('Reconstruction of input sequence')
        for i in range(n_panels):
            axs[i].plot(r['data'][i], color='k', label='Sequence Data')
            axs[i].plot(r['data_tilde'][i], color='r', linestyle='dashed', label='Sequence Reconstruction')
        fig.set_tight_layout(True)
        name = 'Reconstruction_' + model_name + ('_' + suffix if suffix else '') + '.png'
        target = os.path.join(filepath, 'evaluate', name)
        fig.savefig(target, bbox_inches='tight')
    plt.close(fig)
    return r


def plot_loss(cfg, filepath, model_name):
    base = os.path.join(cfg['project_path'], 'model', 'model_losses')
    plt = _pyplot()
    fig, ax = plt.subplots(1, 1)
    fig.suptitle('Losses of our Model')
    ax.set(xlabel='Epochs', ylabel='loss [log-scale]')
    ax.set_yscale('log')
    for stem, label in _LOSS_CURVES:
        ax.plot(np.atleast_1d(np.load(os.path.join(base, stem + model_name + '.npy'))), label=label)
    ax.legend()
    fig.savefig(os.path.join(filepath, 'evaluate', 'MSE-and-KL-Loss' + model_name + '.png'))
    plt.close(fig)


def eval_temporal(cfg, use_gpu, model_name, fixed, snapshot=None, suffix=None):
    T2 = cfg['time_window'] * 2
    F = cfg['num_features'] - (0 if fixed else 2)
    seq_len_half = T2 // 2
    FS = cfg['prediction_steps']
    filepath = os.path.join(cfg['project_path'], 'model')
    dev = _device()
    torch.manual_seed(19)
    model = RNN_VAE(T2, cfg['zdims'], F, cfg['prediction_decoder'], FS, cfg['hidden_size_layer_1'],
                    cfg['hidden_size_layer_2'], cfg['hidden_size_rec'], cfg['hidden_size_pred'], cfg['dropout_encoder'],
                    cfg['dropout_rec'], cfg['dropout_pred'], cfg['softplus'])
    model.engine_options = dict(cfg.get('vame_amd_engine') or {})
    weights = snapshot or os.path.join(filepath, 'best_model', model_name + '_' + cfg['Project'] + '.pkl')
    model.load_state_dict(torch.load(weights, map_location='cpu'))
    model = model.to(dev)
    model.eval()

    testset = SEQUENCE_DATASET(os.path.join(cfg['project_path'], 'data', 'train', ''), data='test_seq.npy', train=False,
                               temporal_window=T2)
    keep = seq_len_half + (FS if cfg['prediction_decoder'] else 0)
    test_loader = DeviceWindowLoader(testset, TEST_BATCH_SIZE, keep, dev)
    r = plot_reconstruction(filepath, test_loader, seq_len_half, model, model_name, cfg['prediction_decoder'], FS,
                            suffix=suffix if snapshot else None)
    plot_loss(cfg, filepath, model_name)
    return r


def evaluate_model(config, use_snapshots=False):
    cfg = read_config(Path(config).resolve())
    model_name = cfg['model_name']
    fixed = cfg['egocentric_data']
    os.makedirs(os.path.join(cfg['project_path'], 'model', 'evaluate'), exist_ok=True)
    use_gpu = torch.cuda.is_available()
    if use_gpu:
        print('GPU used:', torch.cuda.get_device_name(0))
    print("\n\nEvaluation of %s model. \n" % model_name)
    if not use_snapshots:
        eval_temporal(cfg, use_gpu, model_name, fixed)
    else:
        snapdir = os.path.join(cfg['project_path'], 'model', 'best_model', 'snapshots')
        for snap in sorted(os.listdir(snapdir)):
            epoch = snap.split('_')[-1]
            eval_temporal(cfg, use_gpu, model_name, fixed, snapshot=os.path.join(snapdir, snap), suffix='snapshot' + str(epoch))
    print("You can find the results of the evaluation in '%s'" % os.path.join(cfg['project_path'], 'model', 'evaluate'))
