"""Latent embedding + motif parameterisation -- drop-in for vame/analysis/pose_segmentation.py.

Hot path (reference lines): `load_model` :27-64, `embedd_latent_vectors` :67-101 -- the reference
embeds every stride-1 window with batch size 1 and two host<->device copies per window; here the
series is uploaded once, windows are cut by the HIP gather kernel in batches and run through the
encoder + Lambda(mean) kernels; with several ranks the window index range is sharded (no
collective on the data path; one all-gather of the (N-T, Z) result).  Semantics kept: no
mean/std normalisation (:84), windows i in [0, N-T) (:87), eval-mode mu, float32 output (:96-98).
The k-means / HMM parameterisation (:129-191) stays on the host via scikit-learn / hmmlearn.
"""
import os
import pickle
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

from .. import _lib, ops
from ..model.rnn_model import RNN_VAE
from ..util.auxiliary import read_config

EMBED_BATCH = 16384       # windows per launch group: ~1.2 GB of fp32 activations at H=256


def _device():
    return _lib.device()


def load_model(cfg, model_name, fixed):
    NUM_FEATURES = cfg['num_features']
    if fixed == False:  # noqa: E712
        NUM_FEATURES = NUM_FEATURES - 2
    model = RNN_VAE(cfg['time_window'] * 2, cfg['zdims'], NUM_FEATURES, cfg['prediction_decoder'], cfg['prediction_steps'],
                    cfg['hidden_size_layer_1'], cfg['hidden_size_layer_2'], cfg['hidden_size_rec'], cfg['hidden_size_pred'],
                    cfg['dropout_encoder'], cfg['dropout_rec'], cfg['dropout_pred'], cfg['softplus'])
    model.engine_options = dict(cfg.get('vame_amd_engine') or {})
    dev = _device()
    path = os.path.join(cfg['project_path'], 'model', 'best_model', model_name + '_' + cfg['Project'] + '.pkl')
    model.load_state_dict(torch.load(path, map_location="cpu"))
    model = model.to(dev)
    model.eval()
    return model


def embed_series(model, data, batch=EMBED_BATCH, rank=0, world=1):
    """(F,N) array -> (N-T, Z) float32 latents (this rank's shard if world > 1: rows [lo, hi))."""
    eng = model._ensure_engine()
    s, dev = eng.spec, eng.dev
    F, N = data.shape
    if F != s.F:
        raise ValueError(f"series has {F} features, model expects {s.F}")
    n = N - s.T
    lo, hi = (n * rank) // world, (n * (rank + 1)) // world
    # this rank's frames only: windows [lo, hi) read frames [lo, hi + T - 1) (a T-1 frame halo); nothing else crosses PCIe
    view = data[:, lo:max(hi + s.T - 1, lo)]
    Ns = view.shape[1]
    if view.dtype == np.float32 and (Ns == 0 or view.strides[1] == 4):
        X = torch.empty((F, Ns), device=dev)              # float32 rows go up as they lie (no host-side copy of the slice)
        for f in range(F):
            X[f].copy_(torch.from_numpy(view[f]))
    else:
        X = torch.from_numpy(np.ascontiguousarray(view, dtype=np.float32)).to(dev)     # .type(FloatTensor) of :91-94
    out = torch.empty(max(hi - lo, 0), s.Z, device=dev)
    win = torch.empty(min(batch, max(hi - lo, 1)), s.T, F, device=dev)
    with torch.no_grad():
        for i0 in range(lo, hi, batch):
            b = min(batch, hi - i0)
            ops.window_gather(X, Ns, F, None, i0 - lo, b, s.T, win)
            hn = eng.encode(win, s.T * F, b, training=False)
            _, mu, _ = eng.latent(hn, b, None, False, want_kl=False)
            out[i0 - lo:i0 - lo + b].copy_(mu[:b * s.Z].view(b, s.Z))
    eng.check_async_errors()          # small tail batches run the cooperative kernels: never hand out latents of a failed launch
    return out, (lo, hi)


def embedd_latent_vectors(cfg, files, model, fixed):
    project_path = cfg['project_path']
    rank, world = (dist.get_rank(), dist.get_world_size()) if (dist.is_available() and dist.is_initialized()) else (0, 1)
    latent_vector_files = []
    for file in files:
        print('Embedding of latent vector for file %s' % file)
        data = np.load(os.path.join(project_path, 'data', file, file + '-PE-seq-clean.npy'))
        shard, (lo, hi) = embed_series(model, data, rank=rank, world=world)
        if dist.is_available() and dist.is_initialized():
            n = data.shape[1] - cfg['time_window']
            per = -(-n // world)
            padded = torch.zeros(per, shard.shape[1], device=shard.device)
            padded[:hi - lo] = shard
            gathered = [torch.empty_like(padded) for _ in range(world)]
            dist.all_gather(gathered, padded)
            parts = [g[:(n * (r + 1)) // world - (n * r) // world] for r, g in enumerate(gathered)]
            shard = torch.cat(parts, 0)
        latent_vector_files.append(shard.cpu().numpy())
    return latent_vector_files


def get_motif_usage(label):
    """Counts per motif id with zero-filled gaps (pose_segmentation.py:109-126)."""
    ids, counts = np.unique(label, return_counts=True)
    usage = np.zeros(int(ids.max()) - int(ids.min()) + 1, dtype=counts.dtype)
    usage[ids - ids.min()] = counts
    return usage


def _kmeans_cls(cfg):
    """scikit-learn by default (the reference's exact behaviour); cfg['amd_gpu_kmeans']: True runs the same algorithm on
    the MI355X (vame_amd/analysis/kmeans_hip.py) -- not bit-identical to scikit-learn, see that module."""
    if cfg.get('amd_gpu_kmeans', False):
        from .kmeans_hip import KMeansHIP

        def make(init, n_clusters, random_state, n_init):
            return KMeansHIP(n_clusters, n_init=n_init, random_state=random_state)
        return make
    from sklearn.cluster import KMeans
    return KMeans


def _hmm_cls(cfg):
    """hmmlearn's GaussianHMM (the reference's exact behaviour, pose_segmentation.py:20,145-158) unless cfg['amd_gpu_hmm'] is set;
    then the same Baum-Welch / Viterbi on the MI355X (vame_amd/analysis/hmm_hip.py).  The GPU model is never chosen silently: its
    parity with hmmlearn is unpinned and its pickle is not loadable by the reference."""
    if cfg.get('amd_gpu_hmm', False):
        from .hmm_hip import GaussianHMMHIP
        return GaussianHMMHIP
    try:
        from hmmlearn import hmm
    except ImportError as e:
        raise ImportError("parameterization 'hmm' needs hmmlearn (as in the reference); to run the MI355X Gaussian HMM instead set "
                          "amd_gpu_hmm: True in config.yaml (results/hmm_trained.pkl is then a vame_amd model)") from e
    return hmm.GaussianHMM


def same_parameterization(cfg, files, latent_vector_files, states, parameterization):
    KMeans = _kmeans_cls(cfg)
    labels, cluster_centers, motif_usages = [], [], []
    latent_vector_cat = np.concatenate(latent_vector_files, axis=0)
    if parameterization == "kmeans":
        print("Using kmeans as parameterization!")
        kmeans = KMeans(init='k-means++', n_clusters=states, random_state=42, n_init=20).fit(latent_vector_cat)   # :141
        clust_center = kmeans.cluster_centers_
        label = kmeans.predict(latent_vector_cat)
    elif parameterization == "hmm":
        GaussianHMM = _hmm_cls(cfg)
        save_data = os.path.join(cfg['project_path'], "results", "")
        if cfg['hmm_trained'] == False:  # noqa: E712
            print("Using a HMM as parameterization!")
            hmm_model = GaussianHMM(n_components=states, covariance_type="full", n_iter=100)
            hmm_model.fit(latent_vector_cat)
            label = hmm_model.predict(latent_vector_cat)
            with open(save_data + "hmm_trained.pkl", "wb") as file:
                pickle.dump(hmm_model, file)
            with open(save_data + "hmm_trained.impl.txt", "w") as file:         # which implementation wrote the pickle
                file.write(type(hmm_model).__module__ + "." + type(hmm_model).__name__ + "\n")
        else:
            print("Using a pretrained HMM as parameterization!")
            with open(save_data + "hmm_trained.pkl", "rb") as file:
                hmm_model = pickle.load(file)
            label = hmm_model.predict(latent_vector_cat)
    else:
        raise ValueError("parameterization must be 'kmeans' or 'hmm'")
    idx = 0
    for i, file in enumerate(files):
        file_len = latent_vector_files[i].shape[0]
        labels.append(label[idx:idx + file_len])
        if parameterization == "kmeans":
            cluster_centers.append(clust_center)
        motif_usages.append(get_motif_usage(label[idx:idx + file_len]))
        idx += file_len
    return labels, cluster_centers, motif_usages


def individual_parameterization(cfg, files, latent_vector_files, cluster):
    KMeans = _kmeans_cls(cfg)
    random_state = cfg['random_state_kmeans']          # the reference has a KeyError typo here (:175)
    n_init = cfg['n_init_kmeans']
    labels, cluster_centers, motif_usages = [], [], []
    for i, file in enumerate(files):
        print(file)
        kmeans = KMeans(init='k-means++', n_clusters=cluster, random_state=random_state, n_init=n_init).fit(latent_vector_files[i])
        label = kmeans.predict(latent_vector_files[i])
        motif_usages.append(get_motif_usage(label))
        labels.append(label)
        cluster_centers.append(kmeans.cluster_centers_)
    return labels, cluster_centers, motif_usages


def _ask(prompt, rank, world):
    """input() for one process; with several ranks (torchrun: the other ranks have no usable stdin, and different answers would
    leave them in different collectives) rank 0 asks and broadcasts the answer."""
    import builtins
    if world == 1:
        return builtins.input(prompt)
    box = [builtins.input(prompt) if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def _same_on_all_ranks(value, world):
    """Rank 0's view of a file-system test, so that every rank takes the same branch."""
    if world == 1:
        return value
    box = [value]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def pose_segmentation(config):
    config_file = Path(config).resolve()
    cfg = read_config(config_file)
    legacy = cfg['legacy']
    model_name = cfg['model_name']
    n_cluster = cfg['n_cluster']
    fixed = cfg['egocentric_data']
    parameterization = cfg['parameterization']
    print('Pose segmentation for VAME model: %s \n' % model_name)
    if legacy == True:  # noqa: E712
        raise NotImplementedError("pose_segmentation with cfg['legacy']: the reference imports a `segment_behavior` module that is not part "
                                  "of the repository (pose_segmentation.py:205-207), so there is no behaviour to reproduce")
    ind_param = cfg['individual_parameterization']
    pp = cfg['project_path']
    from ..model.rnn_vae import _maybe_init_distributed
    rank, world = _maybe_init_distributed()          # several ranks: the embedding is sharded by window index; rank 0 alone
    is_main = rank == 0                              # parameterises and writes the result files
    for folders in cfg['video_sets']:
        os.makedirs(os.path.join(pp, "results", folders, model_name, ""), exist_ok=True)

    def input(prompt):                               # noqa: A001  several ranks: ask on rank 0 only, every rank gets the answer
        return _ask(prompt, rank, world)

    files = []
    if cfg['all_data'] == 'No':
        all_flag = input("Do you want to qunatify your entire dataset? \n"
                         "If you only want to use a specific dataset type filename: \n"
                         "yes/no/filename ")
        file = all_flag
    else:
        all_flag = 'yes'
    if all_flag in ('yes', 'Yes'):
        files = list(cfg['video_sets'])
        file = files[-1]
    elif all_flag in ('no', 'No'):
        for file in cfg['video_sets']:
            if input("Do you want to quantify " + file + "? yes/no: ") == 'yes':
                files.append(file)
    else:
        files.append(all_flag)

    def res_dir(f):
        return os.path.join(pp, "results", f, model_name, parameterization + '-' + str(n_cluster), "")

    new = True
    if not _same_on_all_ranks(os.path.exists(res_dir(file)), world):
        model = load_model(cfg, model_name, fixed)
        latent_vectors = embedd_latent_vectors(cfg, files, model, fixed)
    else:
        print('\nFor model %s a latent vector embedding already exists. \n'
              'Parameterization of latent vector with %d k-Means cluster' % (model_name, n_cluster))
        flag = input('WARNING: A parameterization for the chosen cluster size of the model already exists! \n'
                     'Do you want to continue? A new parameterization will be computed! (yes/no) ')
        if flag == 'yes':
            latent_vectors = [np.load(os.path.join(res_dir(f), 'latent_vector_' + f + '.npy')) for f in files]
        else:
            print('No new parameterization has been calculated.')
            new = False
    if new and not is_main:
        dist.barrier()                               # rank 0 is writing; nothing to do here
        return
    if new:
        if ind_param == False:  # noqa: E712
            print("For all animals the same parameterization of latent vectors is applied for %d cluster" % n_cluster)
            labels, cluster_center, motif_usages = same_parameterization(cfg, files, latent_vectors, n_cluster, parameterization)
        else:
            print("Individual parameterization of latent vectors for %d cluster" % n_cluster)
            labels, cluster_center, motif_usages = individual_parameterization(cfg, files, latent_vectors, n_cluster)
        for idx, f in enumerate(files):
            save_data = res_dir(f)
            os.makedirs(save_data, exist_ok=True)
            np.save(os.path.join(save_data, str(n_cluster) + '_km_label_' + f), labels[idx])
            if parameterization == "kmeans":
                np.save(os.path.join(save_data, 'cluster_center_' + f), cluster_center[idx])
            np.save(os.path.join(save_data, 'latent_vector_' + f), latent_vectors[idx])
            np.save(os.path.join(save_data, 'motif_usage_' + f), motif_usages[idx])
        if dist.is_available() and dist.is_initialized():
            dist.barrier()
        print("You succesfully extracted motifs with VAME! From here, you can proceed running vame.motif_videos() ")
