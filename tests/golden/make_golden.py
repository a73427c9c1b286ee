#!/usr/bin/env python3
"""Generate golden vectors by importing the REFERENCE implementation (build container only).

Run here (where /root/reference exists):   python tests/golden/make_golden.py
Writes small .npz/.json fixtures next to this file.  Nothing from the reference's source
is stored -- only inputs and the outputs the reference computed for them.

Import recipe: `import vame` fails (ruamel/cv2/umap/hmmlearn absent), so the three
hot-path modules are exec'd by file path under their real dotted names with empty parent
packages and a stub `vame.util.auxiliary`.
"""
import importlib.util
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

REF = os.environ.get("VAME_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    for name in ("vame", "vame.util", "vame.model", "vame.analysis", "hmmlearn"):
        sys.modules.setdefault(name, types.ModuleType(name))
    aux = types.ModuleType("vame.util.auxiliary")
    aux.read_config = lambda path: None
    sys.modules["vame.util.auxiliary"] = aux
    hmm = types.ModuleType("hmmlearn.hmm")
    sys.modules["hmmlearn.hmm"] = hmm
    sys.modules["hmmlearn"].hmm = hmm

    def by_path(dotted, rel):
        spec = importlib.util.spec_from_file_location(dotted, os.path.join(REF, rel))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[dotted] = mod
        spec.loader.exec_module(mod)
        return mod

    rnn_model = by_path("vame.model.rnn_model", "vame/model/rnn_model.py")
    dataloader = by_path("vame.model.dataloader", "vame/model/dataloader.py")
    rnn_vae = by_path("vame.model.rnn_vae", "vame/model/rnn_vae.py")
    pose = by_path("vame.analysis.pose_segmentation", "vame/analysis/pose_segmentation.py")
    return rnn_model, dataloader, rnn_vae, pose


def sd_numpy(model):
    return {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def synth_series(F, N, seed):
    rng = np.random.default_rng(seed)
    n = np.arange(N)
    X = np.sin(2 * np.pi * n[None, :] * (np.arange(F)[:, None] + 1) / 97.0) + 0.5 * rng.standard_normal((F, N))
    return X


def make_model(rnn_model, T, Z, F, future, FS, H, softplus, seed):
    torch.manual_seed(seed)
    return rnn_model.RNN_VAE(2 * T, Z, F, future, FS, H, H, H, H, 0, 0, 0, softplus)


def step_fixture(rnn_model, rnn_vae, name, *, T=30, Z=30, F=24, FS=15, H=32, B=8, future=1, softplus=False,
                 seed=19, kl_weights=(0.0, 0.25, 1.0), adam_steps=3, mse="sum", save_weights=True):
    model = make_model(rnn_model, T, Z, F, future, FS, H, softplus, seed)
    model.train()
    sd0 = sd_numpy(model)
    X = synth_series(F, 400, seed=1)
    Xn = (X - X.mean()) / X.std()
    rs = np.random.RandomState(3)
    starts = rs.randint(0, 400 - 2 * T, size=B)
    full = np.stack([Xn[:, s:s + 2 * T] for s in starts])            # (B,F,2T) f64 like the DataLoader collate
    item = torch.from_numpy(full).permute(0, 2, 1)                     # rnn_vae.py:108
    data = item[:, :T, :].type("torch.FloatTensor")
    fut_t = item[:, T:T + FS, :].type("torch.FloatTensor")
    out = dict(x=data.numpy().copy(), xfut=fut_t.numpy().copy(), starts=starts, series=Xn)
    out["spec"] = np.array([T, F, Z, H, FS, int(future), int(softplus), B])
    if save_weights:
        for k, v in sd0.items():
            out["w/" + k] = v
    else:
        out["w_checksum"] = np.array([float(sum(np.abs(v).astype(np.float64).sum() for v in sd0.values()))])
    # eps capture: randn_like at rnn_model.py:73 is the only RNG draw in forward
    torch.manual_seed(1234)
    eps = torch.randn(B, Z)
    out["eps"] = eps.numpy().copy()
    for w in kl_weights:
        model.zero_grad()
        torch.manual_seed(1234)
        res = model(data)
        if future:
            pred, futp, z, mu, lv = res
        else:
            pred, z, mu, lv = res
            futp = None
        rec = rnn_vae.reconstruction_loss(data, pred, mse)
        kme = rnn_vae.cluster_loss(z.T, Z, 0.1, B)
        kl = rnn_vae.kullback_leibler_loss(mu, lv)
        if future:
            fl = rnn_vae.future_reconstruction_loss(fut_t, futp, mse)
            loss = rec + fl + 1 * w * kl + w * kme
        else:
            fl = torch.zeros(())
            loss = rec + 1 * w * kl + w * kme
        loss.backward()
        tag = f"kw{w:g}/"
        out[tag + "losses"] = np.array([rec.item(), fl.item(), kl.item(), kme.item(), loss.item()], np.float64)
        if w == kl_weights[-1]:
            out["pred"] = pred.detach().numpy().copy()
            if future:
                out["fut"] = futp.detach().numpy().copy()
            out["z"] = z.detach().numpy().copy()
            out["mu"] = mu.detach().numpy().copy()
            out["logvar"] = lv.detach().numpy().copy()
        if save_weights:
            for k, prm in model.named_parameters():
                out[tag + "g/" + k] = prm.grad.detach().numpy().copy()
        else:
            out[tag + "gnorm"] = np.array([prm.grad.norm().item() for _, prm in model.named_parameters()])
            out[tag + "g_l2h"] = model.decoder.latent_to_hidden.weight.grad.numpy().copy()
            out[tag + "g_b_l0"] = model.encoder.encoder_rnn.bias_hh_l0.grad.numpy().copy()
    # eval-mode forward (Lambda returns mean, rnn_model.py:75-76)
    model.eval()
    with torch.no_grad():
        res = model(data)
    out["eval_pred"] = res[0].numpy().copy()
    out["eval_mu"] = res[-2].numpy().copy()
    model.train()
    # Adam-amsgrad trajectory (rnn_vae.py:332,141-143) with kl_weight = 1, fresh eps per step
    if adam_steps and save_weights:
        opt = torch.optim.Adam(model.parameters(), lr=5e-4, amsgrad=True)
        eps_steps, loss_steps = [], []
        for s in range(adam_steps):
            torch.manual_seed(100 + s)
            eps_steps.append(torch.randn(B, Z).numpy().copy())
            torch.manual_seed(100 + s)
            pred, futp, z, mu, lv = model(data)
            loss = (rnn_vae.reconstruction_loss(data, pred, mse) + rnn_vae.future_reconstruction_loss(fut_t, futp, mse)
                    + rnn_vae.kullback_leibler_loss(mu, lv) + rnn_vae.cluster_loss(z.T, Z, 0.1, B))
            opt.zero_grad()
            loss.backward()
            opt.step()
            loss_steps.append(loss.item())
        out["adam/eps"] = np.stack(eps_steps)
        out["adam/loss"] = np.array(loss_steps)
        for k, v in sd_numpy(model).items():
            out["adam/w/" + k] = v
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote", name, {k: round(float(v), 4) for k, v in zip(["rec", "fut", "kl", "km", "tot"], out[f"kw{kl_weights[-1]:g}/losses"])})


def embed_fixture(rnn_model, pose):
    T, Z, F, H = 30, 30, 24, 32
    model = make_model(rnn_model, T, Z, F, 1, 15, H, False, 19)
    model.eval()
    data = synth_series(F, 200, seed=5)
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "data", "vid"))
        np.save(os.path.join(tmp, "data", "vid", "vid-PE-seq-clean.npy"), data)
        cfg = dict(project_path=tmp, time_window=T, num_features=F)
        lat = pose.embedd_latent_vectors(cfg, ["vid"], model, True)[0]
    out = {"w/" + k: v for k, v in sd_numpy(model).items()}
    out.update(data=data, latent=lat, spec=np.array([T, F, Z, H, 15, 1, 0, 0]))
    np.savez_compressed(os.path.join(OUT, "embed_tiny.npz"), **out)
    print("wrote embed_tiny", lat.shape, lat.dtype)


def batcher_fixture(dataloader):
    F, N, T2, B = 24, 500, 60, 16
    X = synth_series(F, N, seed=7) * 3.0 + 1.5
    with tempfile.TemporaryDirectory() as tmp:
        path = tmp + os.sep
        np.save(path + "train_seq.npy", X)
        ds = dataloader.SEQUENCE_DATASET(path, data="train_seq.npy", train=True, temporal_window=T2)
        mean, std = float(np.load(path + "seq_mean.npy")), float(np.load(path + "seq_std.npy"))
        np.random.seed(11)
        items = torch.stack([ds[i] for i in range(B)])               # default collate of B __getitem__ calls
        np.random.seed(11)
        starts = np.random.randint(0, N - T2, size=B)                # stream-equivalent to B scalar np.random.choice
    np.savez_compressed(os.path.join(OUT, "batcher.npz"), X=X, mean=mean, std=std, starts=starts,
                        batch=items.numpy(), T2=T2)
    print("wrote batcher", items.shape, items.dtype)


def h0view_fixture(rnn_model):
    T, Z, F, H = 6, 30, 24, 32
    torch.manual_seed(5)
    dec = rnn_model.Decoder(T, Z, F, H, 0)
    dec.eval()
    out = {"w/decoder." + k: v.detach().numpy().copy() for k, v in dec.state_dict().items()}
    for B in (1, 2, 6):
        torch.manual_seed(B)
        z = torch.randn(B, Z)
        ins = z.unsqueeze(2).repeat(1, 1, T).permute(0, 2, 1)
        with torch.no_grad():
            out[f"B{B}/z"] = z.numpy().copy()
            out[f"B{B}/pred"] = dec(ins, z).numpy().copy()
    out["spec"] = np.array([T, F, Z, H])
    np.savez_compressed(os.path.join(OUT, "h0view.npz"), **out)
    print("wrote h0view")


def decoder_inputs_fixture(rnn_model):
    """Decoder / Decoder_Future run over ARBITRARY `inputs` (rnn_model.py:99-109,132-144: the GRU consumes whatever sequence it is
    given; only RNN_VAE.forward tiles z).  Random inputs that are not z tiled over time."""
    T, Z, F, H, FS = 6, 30, 24, 32, 3
    torch.manual_seed(11)
    dec = rnn_model.Decoder(T, Z, F, H, 0)
    fut = rnn_model.Decoder_Future(T, Z, F, FS, H, 0)
    dec.eval(), fut.eval()
    out = {"w/decoder." + k: v.detach().numpy().copy() for k, v in dec.state_dict().items()}
    out.update({"w/decoder_future." + k: v.detach().numpy().copy() for k, v in fut.state_dict().items()})
    for B in (1, 5):
        torch.manual_seed(20 + B)
        z = torch.randn(B, Z)
        ins = torch.randn(B, T, Z)
        with torch.no_grad():
            out[f"B{B}/z"], out[f"B{B}/ins"] = z.numpy().copy(), ins.numpy().copy()
            out[f"B{B}/pred"] = dec(ins, z).numpy().copy()
            out[f"B{B}/fut"] = fut(ins, z).numpy().copy()            # reads ins[:, :FS] (rnn_model.py:139)
    out["spec"] = np.array([T, F, Z, H, FS])
    np.savez_compressed(os.path.join(OUT, "decoder_inputs.npz"), **out)
    print("wrote decoder_inputs")


def round3_fixtures(rnn_model, rnn_vae):
    """Round 3: mse='mean' pinned at a second KL weight and at H=64 (every gradient of such a step is < 0.1: the checks are relative
    to each tensor's own scale), a hidden size that is not a multiple of 32, and decoders over non-tiled inputs."""
    step_fixture(rnn_model, rnn_vae, "step_tiny_mean_kw025", mse="mean", adam_steps=0, kl_weights=(0.25,))
    step_fixture(rnn_model, rnn_vae, "step_h64_mean", H=64, B=40, T=12, FS=5, adam_steps=0, kl_weights=(0.5,), mse="mean")
    step_fixture(rnn_model, rnn_vae, "step_h40", H=40, B=6, T=8, FS=4, adam_steps=0, kl_weights=(1.0,))
    decoder_inputs_fixture(rnn_model)


def anneal_fixture(rnn_vae):
    tab = {fn: [float(rnn_vae.kl_annealing(e, 2, 4, fn)) for e in range(1, 12)] for fn in ("linear", "sigmoid")}
    with open(os.path.join(OUT, "kl_annealing.json"), "w") as f:
        json.dump(dict(kl_start=2, annealtime=4, epochs=list(range(1, 12)), table=tab), f, indent=1)
    print("wrote kl_annealing")


def train_model_fixture(rnn_vae):
    """Full-driver oracle: run the reference train_model on a synthetic project."""
    import torch.optim.lr_scheduler as lrs

    class _RLROP(lrs.ReduceLROnPlateau):          # torch>=2.4 dropped verbose= (rnn_vae.py:337)
        def __init__(self, *a, verbose=None, **k):
            super().__init__(*a, **k)
    rnn_vae.ReduceLROnPlateau = _RLROP
    F, H = 24, 32
    cfgd = dict(legacy=False, model_name="VAME", pretrained_weights=False, pretrained_model="None", egocentric_data=True,
                Project="demo", batch_size=32, max_epochs=8, zdims=30, beta=1, model_snapshot=3, learning_rate=5e-4,
                num_features=F, time_window=30, prediction_decoder=1, prediction_steps=15, hidden_size_layer_1=H,
                hidden_size_layer_2=H, hidden_size_rec=H, hidden_size_pred=H, dropout_encoder=0, dropout_rec=0,
                dropout_pred=0, noise=False, scheduler_step_size=100, softplus=False, mse_reconstruction_reduction="sum",
                mse_prediction_reduction="sum", kmeans_loss=30, kmeans_lambda=0.1, kl_start=2, annealtime=4,
                anneal_function="linear", scheduler=1, scheduler_gamma=0.2, model_convergence=50)
    train = synth_series(F, 260, seed=21)
    test = synth_series(F, 140, seed=22)
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "data", "train"))
        os.makedirs(os.path.join(tmp, "model"))
        np.save(os.path.join(tmp, "data", "train", "train_seq.npy"), train)
        np.save(os.path.join(tmp, "data", "train", "test_seq.npy"), test)
        cfgd["project_path"] = tmp
        rnn_vae.read_config = lambda p: dict(cfgd)
        with open(os.path.join(tmp, "config.yaml"), "w") as f:
            f.write("x: 1\n")
        np.random.seed(0)
        rnn_vae.train_model(os.path.join(tmp, "config.yaml"))
        ld = os.path.join(tmp, "model", "model_losses")
        out = {n[:-4]: np.load(os.path.join(ld, n)) for n in sorted(os.listdir(ld))}
        out["files_best"] = np.array(sorted(os.listdir(os.path.join(tmp, "model", "best_model"))))
        out["files_snap"] = np.array(sorted(os.listdir(os.path.join(tmp, "model", "best_model", "snapshots"))))
        sd = torch.load(os.path.join(tmp, "model", "best_model", "VAME_demo.pkl"))
        out["sd_keys"] = np.array(list(sd.keys()))
        out["sd_shapes"] = np.array([str(tuple(v.shape)) for v in sd.values()])
    cfgd.pop("project_path")
    out["cfg_json"] = np.array(json.dumps(cfgd))
    out["train_seq"] = train
    out["test_seq"] = test
    np.savez_compressed(os.path.join(OUT, "train_model_run.npz"), **out)
    print("wrote train_model_run", {k: v.shape for k, v in out.items() if k.endswith("VAME")})


def synth_pose_file(F, N, seed, anchors):
    """Pose series with a few gross tracking errors (what the IQR rule removes); two constant rows when `anchors`."""
    rng = np.random.default_rng(seed)
    n = np.arange(N)
    X = 5 * np.sin(2 * np.pi * n[None, :] * (np.arange(F)[:, None] + 1) / 97.0) + rng.standard_normal((F, N))
    spikes = rng.integers(0, N, size=max(3, N // 40))
    fs = rng.integers(0, F, size=len(spikes))
    X[fs, spikes] += rng.choice([-1, 1], size=len(spikes)) * 60
    if anchors:
        X[3] = 0.0
        X[7] = 0.0
    return X


def legacy_fixture(rnn_model, rnn_vae):
    """RNN_VAE_LEGACY (cfg['legacy'], rnn_model.py:186-324): one train step of the reference on a tiny model -- outputs, the
    four loss terms, every gradient (hidden_to_linear gets none), and the eval-mode outputs."""
    T, Z, F, FS, H, B = 30, 30, 24, 15, 32, 6
    torch.manual_seed(23)
    model = rnn_model.RNN_VAE_LEGACY(2 * T, Z, F, 1, FS, H, H, H, H, 0, 0, 0, False)
    model.train()
    out = {"w/" + k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    X = synth_series(F, 400, seed=4)
    Xn = (X - X.mean()) / X.std()
    starts = np.random.RandomState(5).randint(0, 400 - 2 * T, size=B)
    item = torch.from_numpy(np.stack([Xn[:, s:s + 2 * T] for s in starts])).permute(0, 2, 1)
    data = item[:, :T, :].type("torch.FloatTensor")
    fut_t = item[:, T:T + FS, :].type("torch.FloatTensor")
    torch.manual_seed(7)
    eps = torch.randn(B, Z)
    torch.manual_seed(7)                                  # Lambda_LEGACY draws std.data.new(...).normal_(): same stream
    pred, fut, latent, mu, logvar = model(data)
    kw = 0.5
    rec = rnn_vae.reconstruction_loss(data, pred, "sum")
    fl = rnn_vae.future_reconstruction_loss(fut_t, fut, "sum")
    km = rnn_vae.cluster_loss(latent.T, Z, 0.1, B)
    kl = rnn_vae.kullback_leibler_loss(mu, logvar)
    loss = rec + fl + 1.0 * kw * kl + kw * km
    loss.backward()
    np.testing.assert_allclose(latent.detach().numpy(), (eps * torch.exp(0.5 * logvar) + mu).detach().numpy(), atol=1e-6)
    out.update(x=data.numpy().copy(), xfut=fut_t.numpy().copy(), eps=eps.numpy(), pred=pred.detach().numpy(), fut=fut.detach().numpy(),
               z=latent.detach().numpy(), mu=mu.detach().numpy(), logvar=logvar.detach().numpy(),
               losses=np.array([rec.item(), fl.item(), kl.item(), km.item()]), kw=np.array([kw]),
               spec=np.array([T, F, Z, H, FS, 1, 1, B]))
    for k, p in model.named_parameters():
        out["g/" + k] = p.grad.numpy().copy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
    out["no_grad"] = np.array([k for k, p in model.named_parameters() if p.grad is None])
    model.eval()
    with torch.no_grad():
        ep, ef, ez, emu, elv = model(data)
    out.update(eval_pred=ep.numpy(), eval_fut=ef.numpy(), eval_mu=emu.numpy())
    np.savez_compressed(os.path.join(OUT, "step_legacy.npz"), **out)


def prep_fixture():
    """create_trainset arithmetic: the reference's traindata_aligned / traindata_fixed on two small files."""
    import matplotlib
    matplotlib.use("Agg")
    spec = importlib.util.spec_from_file_location("vame.model.create_training", os.path.join(REF, "vame/model/create_training.py"))
    ct = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ct)
    for fixed in (False, True):
        F = 24 if fixed else 26
        datas = [synth_pose_file(F, 300, 1, not fixed), synth_pose_file(F, 220, 2, not fixed)]
        with tempfile.TemporaryDirectory() as d:
            files = ["vidA", "vidB"]
            os.makedirs(os.path.join(d, "data", "train"))
            for f, x in zip(files, datas):
                os.makedirs(os.path.join(d, "data", f))
                np.save(os.path.join(d, "data", f, f + "-PE-seq.npy"), x)
            cfg = dict(project_path=d, robust=True, iqr_factor=4, savgol_length=5, savgol_order=2)
            (ct.traindata_fixed if fixed else ct.traindata_aligned)(cfg, files, 0.1, 26, True, False)
            out = dict(in0=datas[0], in1=datas[1],
                       train=np.load(os.path.join(d, "data", "train", "train_seq.npy")),
                       test=np.load(os.path.join(d, "data", "train", "test_seq.npy")),
                       clean0=np.load(os.path.join(d, "data", "vidA", "vidA-PE-seq-clean.npy")),
                       clean1=np.load(os.path.join(d, "data", "vidB", "vidB-PE-seq-clean.npy")),
                       params=np.array([4, 5, 2, 0.1]))
        np.savez_compressed(os.path.join(OUT, "prep_fixed.npz" if fixed else "prep_aligned.npz"), **out)


def options_fixture(rnn_model, rnn_vae):
    """Model options a stock config.yaml leaves at their defaults: encoder inter-layer dropout (dropout_encoder > 0,
    rnn_model.py:31-35) and decoder hidden sizes that differ from the encoder's (hidden_size_rec / hidden_size_pred,
    rnn_model.py:148-160).  One train step of the reference each: outputs, loss terms, every gradient.
    The dropout keep-mask is reproduced from the torch CPU stream: nn.GRU draws `empty(T,B,2H).bernoulli_(1-p)` between its
    layers (time-major even with batch_first), then Lambda draws randn_like -- checked below against the reference's own z."""
    T, Z, F, FS, B = 30, 30, 24, 15, 8
    X = synth_series(F, 400, seed=1)
    Xn = (X - X.mean()) / X.std()
    starts = np.random.RandomState(3).randint(0, 400 - 2 * T, size=B)
    item = torch.from_numpy(np.stack([Xn[:, s:s + 2 * T] for s in starts])).permute(0, 2, 1)
    data = item[:, :T, :].type("torch.FloatTensor")
    fut_t = item[:, T:T + FS, :].type("torch.FloatTensor")
    for name, (h1, h2, hrec, hpred, pdrop) in (("step_tiny_dropout", (32, 32, 32, 32, 0.25)), ("step_tiny_hsizes", (32, 64, 64, 96, 0.0))):
        torch.manual_seed(19)
        model = rnn_model.RNN_VAE(2 * T, Z, F, 1, FS, h1, h2, hrec, hpred, pdrop, 0, 0, False)
        model.train()
        out = {"w/" + k: v for k, v in sd_numpy(model).items()}
        torch.manual_seed(4321)
        if pdrop > 0:
            mask_tb = torch.empty(T, B, 2 * h1).bernoulli_(1 - pdrop)
            out["drop_mask"] = mask_tb.permute(1, 0, 2).contiguous().numpy()          # (B, T, 2H), keep = 1
        eps = torch.randn(B, Z)
        torch.manual_seed(4321)
        pred, futp, z, mu, lv = model(data)
        np.testing.assert_allclose(z.detach().numpy(), (eps * torch.exp(0.5 * lv) + mu).detach().numpy(), atol=1e-6)
        kw = 0.5
        rec = rnn_vae.reconstruction_loss(data, pred, "sum")
        fl = rnn_vae.future_reconstruction_loss(fut_t, futp, "sum")
        km = rnn_vae.cluster_loss(z.T, Z, 0.1, B)
        kl = rnn_vae.kullback_leibler_loss(mu, lv)
        (rec + fl + 1.0 * kw * kl + kw * km).backward()
        out.update(x=data.numpy().copy(), xfut=fut_t.numpy().copy(), eps=eps.numpy(), pred=pred.detach().numpy(), fut=futp.detach().numpy(),
                   z=z.detach().numpy(), mu=mu.detach().numpy(), logvar=lv.detach().numpy(),
                   losses=np.array([rec.item(), fl.item(), kl.item(), km.item()]), kw=np.array([kw]),
                   spec=np.array([T, F, Z, h1, FS, 1, 0, B, h2, hrec, hpred]), dropout=np.array([pdrop]))
        for k, prm in model.named_parameters():
            out["g/" + k] = prm.grad.numpy().copy()
        model.eval()
        with torch.no_grad():
            ep, ef, ez, emu, elv = model(data)
        out.update(eval_pred=ep.numpy(), eval_fut=ef.numpy(), eval_mu=emu.numpy())
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        print("wrote", name, out["losses"])


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "options":
        rnn_model, dataloader, rnn_vae, pose = load_reference()
        options_fixture(rnn_model, rnn_vae)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "prep":
        load_reference()
        prep_fixture()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "round3":
        rnn_model, dataloader, rnn_vae, pose = load_reference()
        torch.set_num_threads(4)
        round3_fixtures(rnn_model, rnn_vae)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "legacy":
        rnn_model, dataloader, rnn_vae, pose = load_reference()
        legacy_fixture(rnn_model, rnn_vae)
        return
    rnn_model, dataloader, rnn_vae, pose = load_reference()
    torch.set_num_threads(4)
    step_fixture(rnn_model, rnn_vae, "step_tiny")                                       # H=32,B=8, all grads + Adam
    step_fixture(rnn_model, rnn_vae, "step_tiny_oddB", B=5, adam_steps=0, kl_weights=(1.0,))
    step_fixture(rnn_model, rnn_vae, "step_tiny_nofut", future=0, adam_steps=0, kl_weights=(1.0,))
    step_fixture(rnn_model, rnn_vae, "step_tiny_softplus", softplus=True, adam_steps=0, kl_weights=(1.0,))
    step_fixture(rnn_model, rnn_vae, "step_tiny_mean", mse="mean", adam_steps=0, kl_weights=(1.0,))
    step_fixture(rnn_model, rnn_vae, "step_h64", H=64, B=40, T=12, FS=5, adam_steps=0, kl_weights=(0.5,))
    # cfg-size (H=256): weights regenerated from torch.manual_seed(19) on the test box, guarded by checksum
    step_fixture(rnn_model, rnn_vae, "step_cfg256", H=256, B=64, adam_steps=0, kl_weights=(1.0,), save_weights=False)
    embed_fixture(rnn_model, pose)
    batcher_fixture(dataloader)
    h0view_fixture(rnn_model)
    anneal_fixture(rnn_vae)
    train_model_fixture(rnn_vae)
    legacy_fixture(rnn_model, rnn_vae)
    options_fixture(rnn_model, rnn_vae)
    prep_fixture()
    round3_fixtures(rnn_model, rnn_vae)


if __name__ == "__main__":
    main()
