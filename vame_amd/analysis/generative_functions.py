"""Generative sampling through the reconstruction decoder -- counterpart of vame/analysis/generative_functions.py
(SURVEY §8f row N2).

All four modes of the reference end in the same device computation: a handful of latent vectors (GMM samples
:22-70, random embedded windows :73-90, or the k-means cluster centres :93-114) are tiled over `time_window`
steps and pushed through `model.decoder(inputs, z)` (rnn_model.py:99-109), i.e. `latent_to_hidden` with the
`.view(2, B, H)` initial-state mixing, the bidirectional GRU and `hidden_to_output`.  `decode_latents` is that
computation on the HIP kernels; the rest is scikit-learn (GaussianMixture, as in the reference) and matplotlib.
Because of the `.view` mixing the trajectories depend on the *order and number* of latents in the call, so the
callers below pass exactly the sets the reference passes (10 samples / all centres).
"""
import os
from pathlib import Path

import numpy as np
import torch

from ..util.auxiliary import read_config
from .pose_segmentation import load_model as _load_model_fixed

N_GMM_COMPONENTS = 10       # generative_functions.py:30,52
N_SAMPLES = 10              # :33,55,77


def decode_latents(model, latents, time_window):
    """(n, Z) array-like -> (n, time_window, F) float32 reconstructions from `model.decoder` (eval mode)."""
    dev = next(model.parameters()).device
    z = torch.as_tensor(np.asarray(latents), dtype=torch.float32).to(dev).contiguous()
    inputs = z.unsqueeze(2).repeat(1, 1, time_window).permute(0, 2, 1)
    with torch.no_grad():
        traj = model.decoder(inputs, z)
    return traj.detach().cpu().numpy()


def _grid(recon, rows, cols, title, titles=None):
    from ..model.evaluate import _pyplot
    plt = _pyplot()
    fig, axs = plt.subplots(rows, cols, squeeze=False)
    for k in range(min(rows * cols, recon.shape[0])):
        ax = axs[k // cols, k % cols]
        ax.plot(recon[k])
        if titles:
            ax.set_title(titles % k)
    if title:
        fig.suptitle(title)
    return fig


def random_generative_samples_motif(cfg, model, latent_vector, labels, n_cluster):
    """One GMM per motif over that motif's latents, 10 decoded samples each (generative_functions.py:22-47).  Returns the
    decoded samples (n_cluster, N_SAMPLES, T, F)."""
    out = []
    for j in range(n_cluster):
        motif_latents = latent_vector[np.where(labels == j)[0], :]
        gm = _gmm().fit(motif_latents)
        recon = decode_latents(model, gm.sample(N_SAMPLES)[0], cfg['time_window'])
        _grid(recon, 2, 5, 'Generated samples for motif ' + str(j))
        out.append(recon)
    return np.stack(out)


def random_generative_samples(cfg, model, latent_vector):
    gm = _gmm().fit(latent_vector)
    recon = decode_latents(model, gm.sample(N_SAMPLES)[0], cfg['time_window'])
    _grid(recon, 2, 5, 'Generated samples')
    return recon


def random_reconstruction_samples(cfg, model, latent_vector):
    rnd = np.random.choice(latent_vector.shape[0], N_SAMPLES)
    recon = decode_latents(model, latent_vector[rnd], cfg['time_window'])
    _grid(recon, 2, 5, 'Reconstructed samples')
    return recon


def visualize_cluster_center(cfg, model, cluster_center):
    recon = decode_latents(model, cluster_center, cfg['time_window'])
    cols = int(np.ceil(cluster_center.shape[0] / 5))
    _grid(recon, 5, cols, None, titles="Cluster %d")
    return recon


def _gmm():
    from sklearn.mixture import GaussianMixture
    return GaussianMixture(n_components=N_GMM_COMPONENTS)


def load_model(cfg, model_name):
    """The reference's generative loader always drops the two alignment columns (generative_functions.py:122-123)."""
    return _load_model_fixed(cfg, model_name, fixed=False)


def _select_files(cfg):
    if cfg['all_data'] == 'No':
        all_flag = input("Do you want to write motif videos for your entire dataset? \n"
                         "If you only want to use a specific dataset type filename: \n"
                         "yes/no/filename ")
    else:
        all_flag = 'yes'
    if all_flag in ('yes', 'Yes'):
        return list(cfg['video_sets'])
    if all_flag in ('no', 'No'):
        return [f for f in cfg['video_sets'] if input("Do you want to quantify " + f + "? yes/no: ") == 'yes']
    return [all_flag]


MODES = ("sampling", "reconstruction", "centers", "motifs")


def generative_model(config, mode="sampling"):
    if mode not in MODES:
        raise ValueError(f"generative_model: mode must be one of {MODES}, got {mode!r}")
    cfg = read_config(Path(config).resolve())
    model_name, n_cluster = cfg['model_name'], cfg['n_cluster']
    files = _select_files(cfg)
    model = load_model(cfg, model_name)
    results = {}
    for file in files:
        path_to_file = os.path.join(cfg['project_path'], "results", file, model_name, 'kmeans-' + str(n_cluster), "")
        if mode == "sampling":
            results[file] = random_generative_samples(cfg, model, np.load(os.path.join(path_to_file, 'latent_vector_' + file + '.npy')))
        if mode == "reconstruction":
            results[file] = random_reconstruction_samples(cfg, model, np.load(os.path.join(path_to_file, 'latent_vector_' + file + '.npy')))
        if mode == "centers":
            results[file] = visualize_cluster_center(cfg, model, np.load(os.path.join(path_to_file, 'cluster_center_' + file + '.npy')))
        if mode == "motifs":                                                                    # generative_functions.py:191-194
            latent_vector = np.load(os.path.join(path_to_file, 'latent_vector_' + file + '.npy'))
            labels = np.load(os.path.join(path_to_file, str(n_cluster) + '_km_label_' + file + '.npy'))
            results[file] = random_generative_samples_motif(cfg, model, latent_vector, labels, n_cluster)
    return results
