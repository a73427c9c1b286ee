#!/usr/bin/env python3
"""End-to-end demo on a synthetic VAME project (the hot-path part of examples/demo.py of the reference): builds
<tmp>/config.yaml + data/train/{train,test}_seq.npy + data/<video>/<video>-PE-seq-clean.npy, then runs
vame.train_model(config) and vame.pose_segmentation(config) on the MI355X and prints timings."""
import os, sys, time, tempfile, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, yaml
import vame_amd as vame

def synth(F, N, seed):
    rng = np.random.default_rng(seed)
    n = np.arange(N)
    return np.sin(2 * np.pi * n[None, :] * (np.arange(F)[:, None] + 1) / 997.0) + 0.5 * rng.standard_normal((F, N))

def main(epochs=6, batch=256, n_train=200_000, n_test=30_000, n_video=100_000, gpu_kmeans=True, train_only=False):
    root = tempfile.mkdtemp(prefix="vame_demo_")
    F = 24
    cfg = dict(Project="demo", project_path=root, model_name="VAME", legacy=False, pretrained_weights=False, pretrained_model="None",
               egocentric_data=True, batch_size=batch, max_epochs=epochs + 1, zdims=30, beta=1, model_snapshot=50, learning_rate=5e-4,
               num_features=F, time_window=30, prediction_decoder=1, prediction_steps=15, hidden_size_layer_1=256,
               hidden_size_layer_2=256, hidden_size_rec=256, hidden_size_pred=256, dropout_encoder=0, dropout_rec=0, dropout_pred=0,
               noise=False, scheduler_step_size=100, softplus=False, mse_reconstruction_reduction="sum",
               mse_prediction_reduction="sum", kmeans_loss=30, kmeans_lambda=0.1, kl_start=2, annealtime=4, anneal_function="linear",
               scheduler=1, scheduler_gamma=0.2, model_convergence=50, n_cluster=15, parameterization="kmeans",
               individual_parameterization=False, video_sets=["video-1"], all_data="yes", hmm_trained=False,
               random_state_kmeans=42, n_init_kmeans=15, amd_gpu_kmeans=gpu_kmeans)
    os.makedirs(os.path.join(root, "data", "train")); os.makedirs(os.path.join(root, "data", "video-1"))
    os.makedirs(os.path.join(root, "model")); os.makedirs(os.path.join(root, "results", "video-1"))
    np.save(os.path.join(root, "data", "train", "train_seq.npy"), synth(F, n_train, 1))
    np.save(os.path.join(root, "data", "train", "test_seq.npy"), synth(F, n_test, 2))
    np.save(os.path.join(root, "data", "video-1", "video-1-PE-seq-clean.npy"), synth(F, n_video, 3))
    with open(os.path.join(root, "config.yaml"), "w") as f:
        yaml.safe_dump(cfg, f)
    np.random.seed(0)
    t0 = time.perf_counter(); vame.train_model(os.path.join(root, "config.yaml")); t_train = time.perf_counter() - t0
    losses = np.load(os.path.join(root, "model", "model_losses", "mse_train_losses_VAME.npy"))
    if train_only:                       # (kernel-name traces of train_model() alone: the GPU k-means of pose_segmentation() is built from torch ops)
        print("DEMO " + json.dumps(dict(project=root, epochs=epochs, batch=batch, train_seconds=round(t_train, 2), steps=epochs * (n_train // batch),
                                        mse_first=float(losses[0]), mse_last=float(losses[-1]))))
        return
    t0 = time.perf_counter(); vame.pose_segmentation(os.path.join(root, "config.yaml")); t_seg = time.perf_counter() - t0
    lat = np.load(os.path.join(root, "results", "video-1", "VAME", "kmeans-15", "latent_vector_video-1.npy"))
    steps = epochs * (n_train // batch)
    print("DEMO " + json.dumps(dict(project=root, epochs=epochs, batch=batch, train_seconds=round(t_train, 2), steps=steps,
                                    train_windows_per_s=round(steps * batch / t_train), pose_segmentation_seconds=round(t_seg, 2),
                                    latents=list(lat.shape), mse_first=float(losses[0]), mse_last=float(losses[-1]))))

if __name__ == "__main__":
    main(batch=int(sys.argv[1]) if len(sys.argv) > 1 else 256, train_only="train-only" in sys.argv[2:])
