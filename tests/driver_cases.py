"""End-to-end checks of vame.train_model() / vame.pose_segmentation() (and the next-row drivers) on a synthetic project (config 1
shape of BASELINE.json, tiny hidden size): file layout, loss arrays, checkpoint format, the embedding output contract -- compared
with the run of the REFERENCE driver captured in tests/golden/train_model_run.npz.  Shared by test_train_driver_emu.py (host
emulator) and test_train_driver_gpu.py (MI355X, -m gpu)."""
import json
import os

import numpy as np
import pytest
import torch
import yaml

from conftest import load_golden


def _vame():
    """`import vame` exactly as the reference's scripts write it (examples/demo.py:8,48,56), resolved to vame_amd by the opt-in
    alias (vame_amd/compat.py): every driver check below runs the reference's call order through the reference's names."""
    from vame_amd import compat
    compat.install_alias()
    import vame
    assert compat.is_alias(vame)
    return vame


def check_alias_surface():
    """The dotted imports the reference's own modules use for the hot path resolve to the build's classes."""
    import vame_amd
    vame = _vame()
    from vame.model.rnn_vae import RNN_VAE as from_vae                    # vame/model/evaluate.py:20
    from vame.model.rnn_model import RNN_VAE as from_model                # vame/analysis/pose_segmentation.py:24
    from vame.model import SEQUENCE_DATASET, train_model                  # vame/model/__init__.py:16-17
    from vame.analysis import pose_segmentation                           # vame/analysis/__init__.py:14
    from vame.analysis.pose_segmentation import embedd_latent_vectors, load_model          # noqa: F401
    from vame.util.auxiliary import read_config                           # noqa: F401
    assert from_vae is from_model is vame_amd.model.rnn_model.RNN_VAE
    assert train_model is vame.train_model is vame_amd.train_model and pose_segmentation is vame.pose_segmentation
    assert SEQUENCE_DATASET is vame_amd.model.dataloader.SEQUENCE_DATASET
    with pytest.raises(AttributeError, match="not part of the MI355X"):
        vame.motif_videos


def make_project(tmp_path_factory):
    g = load_golden("train_model_run")
    cfg = json.loads(str(g["cfg_json"]))
    root = tmp_path_factory.mktemp("proj")
    os.makedirs(root / "data" / "train")
    os.makedirs(root / "model")
    np.save(root / "data" / "train" / "train_seq.npy", g["train_seq"])
    np.save(root / "data" / "train" / "test_seq.npy", g["test_seq"])
    cfg.update(project_path=str(root), n_cluster=4, parameterization="kmeans", individual_parameterization=False,
               video_sets=["vid1"], all_data="yes", hmm_trained=False, random_state_kmeans=42, n_init_kmeans=3)
    os.makedirs(root / "data" / "vid1")
    np.save(root / "data" / "vid1" / "vid1-PE-seq-clean.npy", g["train_seq"][:, :120])
    os.makedirs(root / "results" / "vid1")
    with open(root / "config.yaml", "w") as f:
        yaml.safe_dump(cfg, f)
    return root, cfg, g


def check_train_model_files_and_losses(project):
    vame = _vame()
    root, cfg, g = project
    np.random.seed(0)
    vame.train_model(str(root / "config.yaml"))
    ld = root / "model" / "model_losses"
    names = sorted(os.listdir(ld))
    assert names == sorted(k + ".npy" for k in g if k.endswith("_VAME"))
    for n in names:
        mine, ref = np.load(ld / n), g[n[:-4]]
        assert mine.shape == ref.shape, n
    # same annealing schedule; first-epoch losses are the untrained model on statistically identical batches
    np.testing.assert_allclose(np.load(ld / "weight_values_VAME.npy"), g["weight_values_VAME"])
    mine, ref = np.load(ld / "mse_train_losses_VAME.npy"), g["mse_train_losses_VAME"]
    assert abs(mine[0] - ref[0]) / ref[0] < 0.1
    assert mine[-1] < mine[0]                                   # it learns
    assert sorted(os.listdir(root / "model" / "best_model")) == list(g["files_best"])
    assert sorted(os.listdir(root / "model" / "best_model" / "snapshots")) == list(g["files_snap"])
    sd = torch.load(root / "model" / "best_model" / "VAME_demo.pkl", map_location="cpu")
    assert list(sd.keys()) == list(g["sd_keys"])
    assert [str(tuple(v.shape)) for v in sd.values()] == list(g["sd_shapes"])
    assert os.path.exists(root / "data" / "train" / "seq_mean.npy")


def check_pose_segmentation_outputs(project):
    vame = _vame()
    root, cfg, g = project
    vame.pose_segmentation(str(root / "config.yaml"))
    out = root / "results" / "vid1" / "VAME" / "kmeans-4"
    lat = np.load(out / "latent_vector_vid1.npy")
    assert lat.shape == (120 - cfg["time_window"], cfg["zdims"]) and lat.dtype == np.float32     # N-T windows
    lab = np.load(out / "4_km_label_vid1.npy")
    assert lab.shape == (lat.shape[0],) and set(np.unique(lab)) <= set(range(4))
    assert np.load(out / "cluster_center_vid1.npy").shape == (4, cfg["zdims"])
    assert np.load(out / "motif_usage_vid1.npy").sum() == lat.shape[0]
    # the embedding is the eval-mode mean of the trained model on un-normalised windows (pose_segmentation.py:84-96)
    from oracle import vame_oracle as vo
    sd = torch.load(root / "model" / "best_model" / "VAME_demo.pkl", map_location="cpu")
    p = {k: v.numpy() for k, v in sd.items()}
    H = cfg["hidden_size_layer_1"]
    ref = vo.embed_series(p, np.load(root / "data" / "vid1" / "vid1-PE-seq-clean.npy"),
                          vo.Spec(T=30, F=24, Z=30, H=H, FS=15), batch=64)
    np.testing.assert_allclose(lat, ref, atol=2e-5)


def _oracle_params(root, cfg):
    from oracle import vame_oracle as vo
    sd = torch.load(root / "model" / "best_model" / "VAME_demo.pkl", map_location="cpu")
    return vo, {k: v.numpy() for k, v in sd.items()}, vo.Spec(T=30, F=24, Z=30, H=cfg["hidden_size_layer_1"], FS=15)


def check_evaluate_model_outputs(project):
    """vame.evaluate_model (evaluate.py:169-215): PNGs under model/evaluate/ and the plotted numbers = eval-mode forward
    of 64 random z-scored test windows (mu feeds both decoders)."""
    vame = _vame()
    from vame_amd.model import evaluate as ev
    root, cfg, g = project
    vame.evaluate_model(str(root / "config.yaml"))
    made = sorted(os.listdir(root / "model" / "evaluate"))
    assert made == ["Future_Reconstruction.png", "MSE-and-KL-LossVAME.png"]
    assert all(os.path.getsize(root / "model" / "evaluate" / m) > 2000 for m in made)
    from vame_amd.util.auxiliary import read_config
    np.random.seed(5)
    r = ev.eval_temporal(read_config(str(root / "config.yaml")), False, "VAME", cfg["egocentric_data"])
    assert r["data"].shape == (64, 30, 24) and r["fut"].shape == (64, 15, 24)
    # the windows are the reference batcher's: z-scored with the TRAIN mean/std, starts from the global numpy stream
    X = np.load(root / "data" / "train" / "test_seq.npy")
    m, sdv = np.load(root / "data" / "train" / "seq_mean.npy"), np.load(root / "data" / "train" / "seq_std.npy")
    np.random.seed(5)
    st = np.random.randint(0, X.shape[1] - 60, size=64)
    win = np.stack([((X[:, s:s + 60] - m) / sdv).T for s in st]).astype(np.float32)
    np.testing.assert_array_equal(r["data"], win[:, :30])
    np.testing.assert_array_equal(r["fut_orig"], win[:, 30:45])
    vo, p, spec = _oracle_params(root, cfg)
    pred, fut, z, mu, lv = vo.model_forward(p, win[:, :30], None, spec, training=False)
    np.testing.assert_allclose(r["data_tilde"], pred, atol=2e-5)
    np.testing.assert_allclose(r["fut"], fut, atol=2e-5)
    np.testing.assert_allclose(r["mu"], mu, atol=2e-5)
    np.testing.assert_array_equal(r["latent"], r["mu"])                     # eval: z = mu (rnn_model.py:75-76)
    # snapshots: one PNG per snapshot file, named like the reference (suffix = 'snapshot' + last '_' token)
    vame.evaluate_model(str(root / "config.yaml"), use_snapshots=True)
    assert len(os.listdir(root / "model" / "evaluate")) == 2               # future-decoder runs overwrite one file


def check_generative_model_modes(project):
    """generative_functions.py: every mode ends in model.decoder(tiled z, z); check against the oracle decoder, incl. the
    h0 .view mixing across the samples of one call (order matters)."""
    from vame_amd.analysis import generative_functions as gf
    root, cfg, g = project
    cfg = dict(cfg, egocentric_data=False, num_features=26)               # the generative loader always drops 2 columns
    import yaml as _y
    with open(root / "config_gen.yaml", "w") as f:
        _y.safe_dump(cfg, f)
    os.replace(root / "config_gen.yaml", root / "config.yaml")
    vo, p, spec = _oracle_params(root, cfg)
    out = root / "results" / "vid1" / "VAME" / "kmeans-4"
    centers = np.load(out / "cluster_center_vid1.npy")
    res = gf.generative_model(str(root / "config.yaml"), mode="centers")["vid1"]
    assert res.shape == (4, 30, 24)
    np.testing.assert_allclose(res, vo.decoder_forward(p, centers.astype(np.float32), 30, "decoder", "rnn_rec"), atol=2e-5)
    lat = np.load(out / "latent_vector_vid1.npy")
    np.random.seed(3)
    res = gf.generative_model(str(root / "config.yaml"), mode="reconstruction")["vid1"]
    np.random.seed(3)
    pick = np.random.choice(lat.shape[0], 10)
    np.testing.assert_allclose(res, vo.decoder_forward(p, lat[pick], 30, "decoder", "rnn_rec"), atol=2e-5)
    res = gf.generative_model(str(root / "config.yaml"), mode="sampling")["vid1"]
    assert res.shape == (10, 30, 24) and np.isfinite(res).all()
    model = gf.load_model(cfg, "VAME")
    perm = gf.decode_latents(model, centers[::-1].copy(), 30)[::-1]
    assert np.abs(perm - gf.decode_latents(model, centers, 30)).max() > 1e-4   # cross-sample h0 mixing is reproduced
    import matplotlib.pyplot as plt
    plt.close("all")


def check_train_model_options(project, tmp_path, capsys):
    """Config keys of the train driver beyond the stock values (SURVEY 8(b) list): `pretrained_weights` (loads best_model/<pretrained_model>_<Project>.pkl
    and forces KL_START = 0, ANNEALTIME = 1: rnn_vae.py:306-323), `noise` (rnn_vae.py:116-119), `scheduler: 0` (StepLR, :339), a non-'sum' reduction, a
    second `model_name`; then the failure branches: a missing pretrained file prints the reference's hint and trains from scratch, an unknown anneal
    function raises NotImplementedError (:75)."""
    import shutil
    vame = _vame()
    root, cfg, g = project
    oroot = tmp_path / "opts"
    shutil.copytree(root / "data", oroot / "data")
    shutil.copytree(root / "model", oroot / "model")                      # best_model/VAME_demo.pkl from the first driver test
    ocfg = dict(cfg, project_path=str(oroot), pretrained_weights=True, pretrained_model="VAME", model_name="VAME2", noise=True, scheduler=0,
                mse_prediction_reduction="mean", max_epochs=4, model_snapshot=2, batch_size=96)
    with open(oroot / "config.yaml", "w") as f:
        yaml.safe_dump(ocfg, f)
    np.random.seed(2)
    vame.train_model(str(oroot / "config.yaml"))
    out = capsys.readouterr().out
    assert "Loading pretrained weights from" in out and "Could not load pretrained model" not in out
    ld = oroot / "model" / "model_losses"
    np.testing.assert_allclose(np.load(ld / "weight_values_VAME2.npy"), [1.0, 1.0, 1.0])          # KL_START = 0, ANNEALTIME = 1 from epoch 1 on
    first = np.load(ld / "mse_train_losses_VAME2.npy")[0]
    assert os.path.exists(oroot / "model" / "best_model" / "VAME2_demo.pkl")                       # weight > 0.99 from the first epoch: saved
    assert sorted(os.listdir(oroot / "model" / "best_model" / "snapshots")) == sorted(list(g["files_snap"]) + ["VAME2_demo_epoch_2.pkl"])
    fut = np.load(ld / "fut_losses_VAME2.npy")
    assert fut[0] < 10.0                                                                           # 'mean' reduction of the future term (sum: ~1e3)
    # a pretrained model that does not exist: the reference prints a hint and goes on with the fresh initialisation
    bad = dict(ocfg, pretrained_model="nope", model_name="VAME3", max_epochs=3)
    with open(oroot / "config_bad.yaml", "w") as f:
        yaml.safe_dump(bad, f)
    os.replace(oroot / "config_bad.yaml", oroot / "config.yaml")
    vame.train_model(str(oroot / "config.yaml"))
    assert "Could not load pretrained model" in capsys.readouterr().out
    np.testing.assert_allclose(np.load(ld / "weight_values_VAME3.npy"), [0.0, 0.0])               # stock kl_start = 2
    assert first < 0.99 * np.load(ld / "mse_train_losses_VAME3.npy")[0]                            # the first run DID start from the trained weights
    with open(oroot / "config.yaml", "w") as f:
        yaml.safe_dump(dict(bad, pretrained_weights=False, anneal_function="cosine", kl_start=0), f)
    with pytest.raises(NotImplementedError):
        vame.train_model(str(oroot / "config.yaml"))
    # model_convergence (rnn_vae.py:375-394): the counter grows while no best model is saved (weight <= 0.99 in the first epochs) and the
    # loop breaks BEFORE that epoch's loss arrays are written, exactly as in the reference
    with open(oroot / "config.yaml", "w") as f:
        yaml.safe_dump(dict(bad, pretrained_weights=False, model_name="VAME4", model_convergence=0, max_epochs=6), f)
    capsys.readouterr()
    vame.train_model(str(oroot / "config.yaml"))
    out = capsys.readouterr().out
    assert "Model converged" in out and out.count("Epoch:") == 1
    assert not os.path.exists(ld / "train_losses_VAME4.npy")


def check_pose_segmentation_prompts(project, tmp_path, monkeypatch):
    """The interactive branches of pose_segmentation() (pose_segmentation.py:218-236,267-277): `all_data: 'No'` asks which files to
    quantify, an existing parameterization asks before recomputing it from the saved latents; `individual_parameterization: True`
    clusters every file on its own with `random_state_kmeans` / `n_init_kmeans`."""
    import builtins
    import shutil
    vame = _vame()
    root, cfg, g = project
    proot = tmp_path / "prompts"
    shutil.copytree(root / "data", proot / "data")
    shutil.copytree(root / "model", proot / "model")
    os.makedirs(proot / "data" / "vid2")
    np.save(proot / "data" / "vid2" / "vid2-PE-seq-clean.npy", g["train_seq"][:, 50:150])
    pcfg = dict(cfg, project_path=str(proot), video_sets=["vid1", "vid2"], all_data="No", individual_parameterization=True, n_cluster=3)
    with open(proot / "config.yaml", "w") as f:
        yaml.safe_dump(pcfg, f)
    asked = []

    def answers(seq):
        it = iter(seq)

        def fake(prompt=""):
            asked.append(prompt)
            return next(it)
        return fake
    monkeypatch.setattr(builtins, "input", answers(["no", "no", "yes"]))        # not everything; vid1: no, vid2: yes
    vame.pose_segmentation(str(proot / "config.yaml"))
    assert len(asked) == 3 and "entire dataset" in asked[0] and "vid1" in asked[1] and "vid2" in asked[2]
    out2 = proot / "results" / "vid2" / "VAME" / "kmeans-3"
    lat = np.load(out2 / "latent_vector_vid2.npy")
    assert lat.shape == (100 - cfg["time_window"], cfg["zdims"]) and np.load(out2 / "cluster_center_vid2.npy").shape == (3, cfg["zdims"])
    assert not os.path.exists(proot / "results" / "vid1" / "VAME" / "kmeans-3")
    lab_first = np.load(out2 / "3_km_label_vid2.npy")
    # second call: the parameterization exists -> asks; 'no' leaves the files alone, 'yes' recomputes from the SAVED latents
    stamp = os.path.getmtime(out2 / "3_km_label_vid2.npy")
    asked.clear()
    monkeypatch.setattr(builtins, "input", answers(["vid2", "no"]))             # a file name instead of yes / no
    vame.pose_segmentation(str(proot / "config.yaml"))
    assert len(asked) == 2 and "already exists" in asked[1] and os.path.getmtime(out2 / "3_km_label_vid2.npy") == stamp
    monkeypatch.setattr(builtins, "input", answers(["vid2", "yes"]))
    vame.pose_segmentation(str(proot / "config.yaml"))
    np.testing.assert_array_equal(np.load(out2 / "latent_vector_vid2.npy"), lat)
    np.testing.assert_array_equal(np.load(out2 / "3_km_label_vid2.npy"), lab_first)               # same latents, same random_state


def check_train_model_legacy_topology(project, tmp_path):
    """cfg['legacy'] = True trains RNN_VAE_LEGACY (rnn_vae.py:294-297): checkpoint keys / shapes of the legacy model."""
    import shutil
    vame = _vame()
    from vame_amd.model.rnn_model import RNN_VAE_LEGACY
    root, cfg, g = project
    lroot = tmp_path / "legacy"
    shutil.copytree(root / "data", lroot / "data")
    os.makedirs(lroot / "model")
    lcfg = dict(cfg, project_path=str(lroot), legacy=True, max_epochs=3, model_snapshot=50, kl_start=0, annealtime=1)
    with open(lroot / "config.yaml", "w") as f:
        yaml.safe_dump(lcfg, f)
    np.random.seed(1)
    vame.train_model(str(lroot / "config.yaml"))
    sd = torch.load(lroot / "model" / "best_model" / "VAME_demo.pkl", map_location="cpu")
    H = cfg["hidden_size_layer_1"]
    ref = RNN_VAE_LEGACY(60, 30, 24, 1, 15, H, H, H, H, 0, 0, 0, False).state_dict()
    assert list(sd.keys()) == list(ref.keys()) and all(sd[k].shape == ref[k].shape for k in sd)
    assert "decoder.rnn_rec.weight_ih_l0_reverse" not in sd and "lmbda.hidden_to_linear.weight" in sd
    losses = np.load(lroot / "model" / "model_losses" / "train_losses_VAME.npy")
    assert losses.shape == (2,) and np.isfinite(losses).all()
    with pytest.raises(NotImplementedError):
        vame.pose_segmentation(str(lroot / "config.yaml"))


def check_create_trainset_files(tmp_path):
    """vame.create_trainset (create_training.py:267-300): train/test split + per-video clean files, equal to the files the
    REFERENCE wrote for the same inputs (tests/golden/prep_*.npz)."""
    vame = _vame()
    for name, fixed in (("prep_aligned", False), ("prep_fixed", True)):
        g = load_golden(name)
        root = tmp_path / name
        for f, k in (("vidA", "in0"), ("vidB", "in1")):
            os.makedirs(root / "data" / f)
            np.save(root / "data" / f / (f + "-PE-seq.npy"), g[k])
        cfg = dict(project_path=str(root), Project="demo", legacy=False, egocentric_data=fixed, all_data="yes", video_sets=["vidA", "vidB"],
                   robust=True, iqr_factor=int(g["params"][0]), savgol_filter=True, savgol_length=int(g["params"][1]),
                   savgol_order=int(g["params"][2]), test_fraction=float(g["params"][3]), num_features=26)
        with open(root / "config.yaml", "w") as f:
            yaml.safe_dump(cfg, f)
        vame.create_trainset(str(root / "config.yaml"))
        np.testing.assert_array_equal(np.load(root / "data" / "train" / "train_seq.npy"), g["train"])
        np.testing.assert_array_equal(np.load(root / "data" / "train" / "test_seq.npy"), g["test"])
        np.testing.assert_array_equal(np.load(root / "data" / "vidA" / "vidA-PE-seq-clean.npy"), g["clean0"])
        np.testing.assert_array_equal(np.load(root / "data" / "vidB" / "vidB-PE-seq-clean.npy"), g["clean1"])
        vame.create_trainset(str(root / "config.yaml"), check_parameter=True)          # plots only, writes nothing new
    import matplotlib.pyplot as plt
    plt.close("all")


def check_read_config_contract(tmp_path):
    from vame_amd.util.auxiliary import read_config
    with pytest.raises(FileNotFoundError):
        read_config(tmp_path / "missing.yaml")
    p = tmp_path / "config.yaml"
    with open(p, "w") as f:
        yaml.safe_dump(dict(project_path="/somewhere/else", Project="x"), f)
    cfg = read_config(str(p))
    assert cfg["project_path"] == str(tmp_path)                 # rewritten when the folder moved (auxiliary.py:139-142)
    assert yaml.safe_load(open(p))["project_path"] == str(tmp_path)


def check_parameterization_with_gpu_kmeans_option():
    """cfg['amd_gpu_kmeans'] routes same_/individual_parameterization through KMeansHIP with the same return contract."""
    from vame_amd.analysis.pose_segmentation import individual_parameterization, same_parameterization
    rng = np.random.default_rng(1)
    cent = rng.standard_normal((4, 30)) * 5
    lat = [(cent[rng.integers(0, 4, n)] + 0.3 * rng.standard_normal((n, 30))).astype(np.float32) for n in (90, 70)]
    cfg = dict(amd_gpu_kmeans=True, random_state_kmeans=42, n_init_kmeans=2, project_path="/tmp")
    for fn, args in ((same_parameterization, (cfg, ["a", "b"], lat, 4, "kmeans")), (individual_parameterization, (cfg, ["a", "b"], lat, 4))):
        labels, centers, usage = fn(*args)
        assert [len(l) for l in labels] == [90, 70] and all(c.shape == (4, 30) for c in centers)
        assert all(u.sum() == n for u, n in zip(usage, (90, 70)))
        ref, _, _ = fn(dict(cfg, amd_gpu_kmeans=False), *args[1:])
        from sklearn.metrics import adjusted_rand_score
        assert all(adjusted_rand_score(a, b) == 1.0 for a, b in zip(labels, ref))


def check_pose_segmentation_hmm(project):
    """cfg['parameterization'] = 'hmm' (pose_segmentation.py:145-158): Gaussian HMM over the latents (GaussianHMMHIP when hmmlearn is absent or
    cfg['amd_gpu_hmm']), labels / motif usage files, results/hmm_trained.pkl, and the hmm_trained = True reload path."""
    import pickle
    vame = _vame()
    root, cfg, g = project
    hcfg = dict(cfg, parameterization="hmm", n_cluster=3, amd_gpu_hmm=True, hmm_trained=False)
    with open(root / "config_hmm.yaml", "w") as f:
        yaml.safe_dump(hcfg, f)
    vame.pose_segmentation(str(root / "config_hmm.yaml"))
    out = root / "results" / "vid1" / "VAME" / "hmm-3"
    lat = np.load(out / "latent_vector_vid1.npy")
    lab = np.load(out / "3_km_label_vid1.npy")
    assert lab.shape == (lat.shape[0],) and set(np.unique(lab)) <= {0, 1, 2}
    assert np.load(out / "motif_usage_vid1.npy").sum() == lat.shape[0]
    assert not os.path.exists(out / "cluster_center_vid1.npy")                      # only the k-means branch writes centres (:306)
    with open(root / "results" / "hmm_trained.pkl", "rb") as f:
        model = pickle.load(f)
    np.testing.assert_array_equal(model.predict(lat), lab)
    # the fitted model is a local optimum of its own likelihood: the oracle's E-step on it reproduces the final log-likelihood trend
    from oracle.hmm_oracle import GaussianHMMOracle
    ref = GaussianHMMOracle(3)
    ref.startprob_, ref.transmat_, ref.means_, ref.covars_ = model.startprob_, model.transmat_, model.means_, model.covars_
    assert ref.score(lat.astype(np.float64)) >= model.history_[-1] - 1e-6 * abs(model.history_[-1])
    np.testing.assert_array_equal(ref.predict(lat), lab)
    # pretrained path: delete the result folder, reload the pickle
    import shutil
    shutil.rmtree(out)
    with open(root / "config_hmm.yaml", "w") as f:
        yaml.safe_dump(dict(hcfg, hmm_trained=True), f)
    vame.pose_segmentation(str(root / "config_hmm.yaml"))
    np.testing.assert_array_equal(np.load(out / "3_km_label_vid1.npy"), lab)
