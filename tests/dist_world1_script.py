"""Run the WHOLE collective path of the package in ONE rank (VAME_AMD_FORCE_DIST=1): process-group init through the backend the
package picks (nccl = RCCL on an MI355X, gloo under the CPU emulator harness), all-reduce of the real flat gradient bucket, one
train_model() run (barriers, rank-averaged statistics) and pose_segmentation() (all-gather of the embedding), group shutdown.
Prints `WORLD1_OK <json>`.  Used by tests/test_distributed_gpu.py (-m gpu) and tests/test_distributed_cpu.py (emulator).

    python tests/dist_world1_script.py <project dir with config.yaml> [emu]
"""
import json
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    root = sys.argv[1]
    emu = len(sys.argv) > 2 and sys.argv[2] == "emu"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VAME_AMD_FORCE_DIST="1")
    import numpy as np
    import torch
    import torch.distributed as dist
    if emu:
        sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
        import harness
        harness.install()
    from conftest import load_golden
    from model_cases import build_model
    import vame_amd as vame
    from vame_amd import _lib
    from vame_amd.analysis.pose_segmentation import embed_series
    from vame_amd.model import rnn_vae
    dev = _lib.device()
    rank, world = rnn_vae._maybe_init_distributed()
    assert (rank, world) == (0, 1) and dist.is_initialized()
    backend = dist.get_backend()
    assert backend == ("gloo" if emu else "nccl"), backend
    # 1. the gradient bucket through the collective: with one rank SUM is the identity, bit for bit, and the status slot stays 0
    g = load_golden("step_h64")
    model, (T, F, Z, H, FS, fut, sp) = build_model(g, dev)
    model.train()
    opt = rnn_vae.FusedAdamAMSGrad(model, lr=5e-4)
    win = torch.cat([torch.from_numpy(g["x"]), torch.from_numpy(g["xfut"])], 1).contiguous().to(dev)
    eps = torch.from_numpy(g["eps"]).to(dev)
    model.loss_step(win, 0.5, beta=1.0, kloss=Z, klmbda=0.1, bsize=win.shape[0], eps=eps)
    g0 = model.flat_parameters()[1].clone()
    gscale = rnn_vae.allreduce_gradients(model)
    assert gscale == 1.0
    assert torch.equal(model.flat_parameters()[1], g0) and float(model._flat_g_comm[-4:].abs().sum()) == 0.0
    for k, prm in model.named_parameters():                   # ... and they are the reference's gradients
        r = g["kw0.5/g/" + k]
        assert np.abs(prm.grad.cpu().numpy() - r).max() <= 5e-5 * np.abs(r).max(), k
    opt.step(gscale=gscale)
    bucket_bytes = model._flat_g_comm.numel() * 4
    # 2. the public drivers under the process group
    np.random.seed(0)
    vame.train_model(os.path.join(root, "config.yaml"))
    losses = np.load(os.path.join(root, "model", "model_losses", "train_losses_VAME.npy"))
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
    vame.pose_segmentation(os.path.join(root, "config.yaml"))
    import yaml
    with open(os.path.join(root, "config.yaml")) as f:
        cfg = yaml.safe_load(f)
    out = os.path.join(root, "results", "vid1", "VAME", "kmeans-%d" % cfg["n_cluster"])
    lat = np.load(os.path.join(out, "latent_vector_vid1.npy"))
    from vame_amd.analysis.pose_segmentation import load_model
    m2 = load_model(cfg, "VAME", cfg["egocentric_data"])
    direct, _ = embed_series(m2, np.load(os.path.join(root, "data", "vid1", "vid1-PE-seq-clean.npy")))
    assert np.array_equal(lat, direct.cpu().numpy())           # the all-gathered shards of one rank = the plain embedding
    rnn_vae.shutdown_distributed()
    assert not dist.is_initialized()
    lib = "gloo"
    if not emu:
        lib = "RCCL " + ".".join(str(v) for v in torch.cuda.nccl.version())
    print("WORLD1_OK " + json.dumps(dict(backend=backend, library=lib, bucket_bytes=bucket_bytes, epochs=len(losses),
                                         device=torch.cuda.get_device_name(0) if not emu else "host emulator")))


if __name__ == "__main__":
    main()
