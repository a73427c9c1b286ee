"""The numpy oracle (oracle/vame_oracle.py) against golden vectors produced by the imported reference."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_weights, load_golden
from tolerances import TINY_REL, assert_grad_close, assert_loss_close, step_scale_of
from oracle import vame_oracle as vo


def spec_of(g):
    T, F, Z, H, FS, fut, sp = [int(v) for v in g["spec"][:7]]
    return vo.Spec(T=T, F=F, Z=Z, H=H, FS=FS, future=bool(fut), softplus=bool(sp))


@pytest.mark.parametrize("name,kw,mse", [
    ("step_tiny", 0.0, "sum"), ("step_tiny", 0.25, "sum"), ("step_tiny", 1.0, "sum"),
    ("step_tiny_oddB", 1.0, "sum"), ("step_tiny_nofut", 1.0, "sum"), ("step_tiny_softplus", 1.0, "sum"),
    ("step_tiny_mean", 1.0, "mean"), ("step_h64", 0.5, "sum"),
    ("step_tiny_mean_kw025", 0.25, "mean"), ("step_h64_mean", 0.5, "mean"), ("step_h40", 1.0, "sum"),
])
def test_step_forward_losses_grads(name, kw, mse):
    g = load_golden(name)
    spec = spec_of(g)
    p = golden_weights(g)
    cache = vo.FwdCache()
    pred, fut, z, mu, lv = vo.model_forward(p, g["x"], g["eps"], spec, True, cache)
    tag = f"kw{kw:g}/"
    if tag + "losses" in g and "pred" in g and kw == max(float(k[2:-7]) for k in g if k.endswith("/losses")):
        np.testing.assert_allclose(pred, g["pred"], atol=2e-5)
        np.testing.assert_allclose(mu, g["mu"], atol=1e-5)
        np.testing.assert_allclose(lv, g["logvar"], atol=1e-5)
        np.testing.assert_allclose(z, g["z"], atol=1e-5)
        if spec.future:
            np.testing.assert_allclose(fut, g["fut"], atol=2e-5)
    L = vo.total_loss(pred, fut, z, mu, lv, g["x"], g["xfut"], spec, kw, mse_red=mse, mse_pred=mse)
    ref = g[tag + "losses"]
    for i, k in enumerate(["rec", "fut", "kl", "kmeans", "total"]):
        assert_loss_close(L[k], ref[i], name=k)
    # the reference's (B,B)-SVD form equals the (Z,Z)-Gram form
    assert_loss_close(vo.cluster_loss_svd(z, spec.Z, 0.1, z.shape[0]), ref[3], name="svd form")
    grads = vo.model_backward(p, cache, spec, g["x"], g["xfut"], kw, mse_red=mse, mse_pred=mse)
    sscale = step_scale_of(g[tag + "g/" + k] for k in grads)
    for k, gv in grads.items():
        assert_grad_close(gv, g[tag + "g/" + k], TINY_REL, k, sscale)            # relative to each tensor's own max


def test_decoders_over_arbitrary_inputs():
    """Decoder / Decoder_Future called with inputs that are NOT z tiled over time (rnn_model.py:99-109,132-144)."""
    g = load_golden("decoder_inputs")
    T, F, Z, H, FS = [int(v) for v in g["spec"]]
    p = golden_weights(g)
    for B in (1, 5):
        z, ins = g[f"B{B}/z"], g[f"B{B}/ins"]
        np.testing.assert_allclose(vo.decoder_forward(p, z, T, "decoder", "rnn_rec", inputs=ins), g[f"B{B}/pred"], atol=2e-6)
        np.testing.assert_allclose(vo.decoder_forward(p, z, FS, "decoder_future", "rnn_pred", inputs=ins), g[f"B{B}/fut"], atol=2e-6)


def test_eval_mode_forward():
    g = load_golden("step_tiny")
    spec = spec_of(g)
    p = golden_weights(g)
    pred, fut, z, mu, lv = vo.model_forward(p, g["x"], None, spec, training=False)
    np.testing.assert_allclose(mu, g["eval_mu"], atol=1e-5)
    np.testing.assert_allclose(pred, g["eval_pred"], atol=2e-5)
    assert z is mu


def test_adam_amsgrad_trajectory():
    g = load_golden("step_tiny")
    spec = spec_of(g)
    p = golden_weights(g)
    st = tuple({k: np.zeros_like(v) for k, v in p.items()} for _ in range(3))
    for s in range(g["adam/eps"].shape[0]):
        cache = vo.FwdCache()
        out = vo.model_forward(p, g["x"], g["adam/eps"][s], spec, True, cache)
        L = vo.total_loss(*out, g["x"], g["xfut"], spec, 1.0)
        assert abs(L["total"] - g["adam/loss"][s]) <= 1e-4 * abs(g["adam/loss"][s])
        grads = vo.model_backward(p, cache, spec, g["x"], g["xfut"], 1.0)
        vo.adam_amsgrad_step(p, grads, st, 5e-4, s + 1)
    for k, v in golden_weights(g, "adam/w/").items():
        np.testing.assert_allclose(p[k], v, atol=2e-5, err_msg=k)


def test_embedding_matches_reference_loop():
    g = load_golden("embed_tiny")
    spec = spec_of(g)
    lat = vo.embed_series(golden_weights(g), g["data"], spec, batch=64)
    assert lat.shape == g["latent"].shape == (g["data"].shape[1] - spec.T, spec.Z) and lat.dtype == np.float32
    np.testing.assert_allclose(lat, g["latent"], atol=1e-5)


def test_window_batcher():
    g = load_golden("batcher")
    Xn = vo.normalise_series(g["X"], float(g["mean"]), float(g["std"]))
    win = vo.window_gather(Xn, g["starts"], int(g["T2"]))          # (B,2T,F)
    np.testing.assert_array_equal(np.transpose(win, (0, 2, 1)), g["batch"])   # bit-exact f64
    np.random.seed(11)
    assert (np.random.randint(0, g["X"].shape[1] - int(g["T2"]), size=len(g["starts"])) == g["starts"]).all()


@pytest.mark.parametrize("B", [1, 2, 6])
def test_decoder_h0_view_mixing(B):
    g = load_golden("h0view")
    T, F, Z, H = [int(v) for v in g["spec"]]
    p = golden_weights(g)
    pred = vo.decoder_forward(p, g[f"B{B}/z"], T, "decoder", "rnn_rec")
    np.testing.assert_allclose(pred, g[f"B{B}/pred"], atol=1e-5)


def test_kl_annealing_table():
    with open(os.path.join(GOLDEN, "kl_annealing.json")) as f:
        t = json.load(f)
    for fn, vals in t["table"].items():
        for e, v in zip(t["epochs"], vals):
            assert vo.kl_annealing(e, t["kl_start"], t["annealtime"], fn) == pytest.approx(v, abs=1e-12)
    with pytest.raises(NotImplementedError):
        vo.kl_annealing(5, 2, 4, "cosine")


def test_torch_cpu_baseline_model_matches_reference():
    """oracle/torch_ref.py (the CPU baseline timed by bench.py) reproduces the reference's losses and grads."""
    import torch
    from oracle.torch_ref import TorchRef, reference_loss
    g = load_golden("step_tiny")
    spec = spec_of(g)
    m = TorchRef(spec.T, spec.F, spec.Z, spec.H, spec.FS, spec.future, spec.softplus)
    m.load_reference_state(golden_weights(g))
    m.train()
    x, xf, eps = [torch.from_numpy(g[k]) for k in ("x", "xfut", "eps")]
    loss, terms = reference_loss(m(x, eps), x, xf, 1.0)
    loss.backward()
    ref = g["kw1/losses"]
    for v, r in zip(list(terms) + [loss], ref):
        assert_loss_close(v.item(), r, rel=1e-5)
    for k, gr in m.reference_named_grads().items():
        r = g["kw1/g/" + k]
        assert_grad_close(gr.numpy(), r, 1e-5, k)


@pytest.mark.parametrize("name,fixed", [("prep_aligned", False), ("prep_fixed", True)])
def test_prep_oracle_matches_reference_outputs(name, fixed):
    """oracle/prep_oracle.py (create_trainset arithmetic incl. the two NaN-fill quirks) vs the reference's saved files: bit-exact."""
    from oracle import prep_oracle as po
    g = load_golden(name)
    f, L, p, frac = g["params"]
    r = po.traindata([g["in0"], g["in1"]], fixed=fixed, robust=True, iqr_factor=int(f), savgol_filter=True, savgol_length=int(L),
                     savgol_order=int(p), test_fraction=float(frac))
    np.testing.assert_array_equal(r["train"], g["train"])
    np.testing.assert_array_equal(r["test"], g["test"])
    np.testing.assert_array_equal(r["clean"][0], g["clean0"])
    np.testing.assert_array_equal(r["clean"][1], g["clean1"])
    assert not np.isnan(g["train"]).any() and g["train"].shape[0] == 24


def test_torch_legacy_restatement_matches_reference():
    """oracle/torch_ref.TorchRefLegacy (stock torch modules) vs one train step of the reference's RNN_VAE_LEGACY."""
    import torch
    from oracle.torch_ref import TorchRefLegacy, reference_loss
    g = load_golden("step_legacy")
    T, F, Z, H, FS, fut, sp, B = [int(v) for v in g["spec"]]
    m = TorchRefLegacy(T, F, Z, H, FS, True)
    m.load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g if k.startswith("w/")})
    m.train()
    x, xfut, eps = [torch.from_numpy(g[k]) for k in ("x", "xfut", "eps")]
    out = m(x, eps)
    loss, terms = reference_loss(out, x, xfut, float(g["kw"][0]), kloss=Z, bsize=B)
    loss.backward()
    np.testing.assert_allclose([float(t) for t in terms], g["losses"], rtol=2e-5)
    np.testing.assert_allclose(out[0].detach().numpy(), g["pred"], atol=2e-6)
    for k, p in m.named_parameters():
        ref = g["g/" + k]
        got = p.grad.numpy() if p.grad is not None else np.zeros_like(ref)
        assert_grad_close(got, ref, TINY_REL, k, step_scale_of(g["g/" + kk] for kk, _ in m.named_parameters()))


@pytest.mark.parametrize("name", ["step_tiny_dropout", "step_tiny_hsizes"])
def test_model_options_dropout_and_hidden_sizes(name):
    """Encoder inter-layer dropout (rnn_model.py:31-35) and decoder hidden sizes != encoder's (:148-160): the oracle against one
    train step of the reference itself (mask reproduced from torch's CPU stream by tests/golden/make_golden.py)."""
    g = load_golden(name)
    T, F, Z, H, FS, fut, sp, B = [int(v) for v in g["spec"][:8]]
    spec = vo.Spec(T=T, F=F, Z=Z, H=H, FS=FS, future=True, softplus=False, dropout=float(g["dropout"][0]))
    p = golden_weights(g)
    cache = vo.FwdCache()
    mask = g["drop_mask"] if "drop_mask" in g else None
    pred, futp, z, mu, lv = vo.model_forward(p, g["x"], g["eps"], spec, True, cache, drop_mask=mask)
    for got, key in ((pred, "pred"), (futp, "fut"), (z, "z"), (mu, "mu"), (lv, "logvar")):
        np.testing.assert_allclose(got, g[key], atol=2e-5, err_msg=key)
    kw = float(g["kw"][0])
    L = vo.total_loss(pred, futp, z, mu, lv, g["x"], g["xfut"], spec, kw)
    for i, k in enumerate(["rec", "fut", "kl", "kmeans"]):
        assert_loss_close(L[k], g["losses"][i], name=k)
    grads = vo.model_backward(p, cache, spec, g["x"], g["xfut"], kw)
    for k, gv in grads.items():
        assert_grad_close(gv, g["g/" + k], TINY_REL, k, step_scale_of(g["g/" + kk] for kk in grads))
    ep, ef, ez, emu, elv = vo.model_forward(p, g["x"], None, spec, training=False, drop_mask=mask)     # eval: no dropout
    np.testing.assert_allclose(emu, g["eval_mu"], atol=1e-5)
    np.testing.assert_allclose(ep, g["eval_pred"], atol=2e-5)


def test_hmm_oracle_vs_sklearn_mixture_in_the_iid_limit():
    """A second, third-party anchor at a realistic length (hmmlearn itself cannot be obtained here): an HMM whose
    transition rows all equal its start distribution w emits i.i.d. draws from the mixture sum_k w_k N(mu_k, S_k), so its
    log-likelihood and state posteriors over 3000 frames must equal scikit-learn's GaussianMixture with the same parameters --
    which exercises the log-domain forward / backward recursions, the density and the posterior normalisation far beyond the
    chain lengths brute-force enumeration can reach."""
    from sklearn.mixture import GaussianMixture
    from oracle import hmm_oracle as ho
    rng = np.random.default_rng(11)
    K, D, N = 4, 5, 3000
    X = rng.standard_normal((N, D)) * rng.uniform(0.5, 2.0, D) + rng.standard_normal(D)
    w = rng.dirichlet(np.ones(K) * 3)
    means = rng.standard_normal((K, D))
    A = rng.standard_normal((K, D, D))
    covars = A @ A.transpose(0, 2, 1) / D + 0.3 * np.eye(D)
    o = ho.GaussianHMMOracle(K)
    o.startprob_, o.transmat_, o.means_, o.covars_ = w, np.tile(w, (K, 1)), means, covars
    gm = GaussianMixture(K, covariance_type="full")
    gm.weights_, gm.means_, gm.covariances_ = w, means, covars
    gm.precisions_cholesky_ = np.stack([np.linalg.cholesky(np.linalg.inv(c)) for c in covars])        # sklearn's internal form: P = L L^T
    logprob, stats, post = o.e_step(X)
    np.testing.assert_allclose(logprob, gm.score_samples(X).sum(), rtol=1e-11)
    np.testing.assert_allclose(post, gm.predict_proba(X), atol=1e-10)
    np.testing.assert_allclose(stats["trans"].sum(0), post[1:].sum(0), atol=1e-8)                        # expected transition counts into each state


@pytest.mark.parametrize("K,N,D,seed", [(2, 6, 2, 0), (3, 5, 1, 1), (2, 7, 3, 2)])
def test_hmm_oracle_vs_exhaustive_enumeration(K, N, D, seed):
    """hmmlearn is not available to pin oracle/hmm_oracle.py against, so its recursions are pinned against the definition instead: on a
    tiny chain every one of the K^N state paths is enumerated -- likelihood, state posteriors, expected transition counts and the best
    path follow by brute force, with no recursion shared with the restatement."""
    import itertools
    from scipy.special import logsumexp
    from oracle import hmm_oracle as ho
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, D))
    o = ho.GaussianHMMOracle(K)
    o.startprob_ = rng.dirichlet(np.ones(K))
    o.transmat_ = rng.dirichlet(np.ones(K), size=K)
    o.means_ = rng.standard_normal((K, D))
    A = rng.standard_normal((K, D, D))
    o.covars_ = A @ A.transpose(0, 2, 1) + 0.5 * np.eye(D)
    logB = ho.log_mvn_density_full(X, o.means_, o.covars_)
    # the density itself against scipy's multivariate normal
    from scipy.stats import multivariate_normal
    for k in range(K):
        np.testing.assert_allclose(logB[:, k], multivariate_normal(o.means_[k], o.covars_[k]).logpdf(X).reshape(N), rtol=1e-10, atol=1e-10)
    paths = list(itertools.product(range(K), repeat=N))
    lp = np.array([np.log(o.startprob_[p[0]]) + logB[0, p[0]] + sum(np.log(o.transmat_[p[t - 1], p[t]]) + logB[t, p[t]] for t in range(1, N))
                   for p in paths])
    logL = logsumexp(lp)
    w = np.exp(lp - logL)                                            # posterior of every path
    gamma = np.zeros((N, K)); xi = np.zeros((K, K))
    for p, wp in zip(paths, w):
        for t in range(N):
            gamma[t, p[t]] += wp
        for t in range(1, N):
            xi[p[t - 1], p[t]] += wp
    logprob, stats, post = o.e_step(X)
    np.testing.assert_allclose(logprob, logL, rtol=1e-12)
    np.testing.assert_allclose(post, gamma, atol=1e-12)
    np.testing.assert_allclose(stats["trans"], xi, atol=1e-12)
    np.testing.assert_allclose(stats["obs"], gamma.T @ X, atol=1e-12)
    np.testing.assert_allclose(o.score(X), logL, rtol=1e-12)
    vlp, vpath = ho.viterbi_log(np.log(o.startprob_), np.log(o.transmat_), logB)
    best = int(np.argmax(lp))
    np.testing.assert_allclose(vlp, lp[best], rtol=1e-12)
    assert tuple(int(v) for v in vpath) == paths[best]
    # one M step: the closed forms with hmmlearn 0.2.8's priors (means_weight 0, covars_prior 1e-2, covars_weight 1 -> max(1 - D, 0) = 0)
    o.m_step(stats)
    np.testing.assert_allclose(o.startprob_, gamma[0] / gamma[0].sum(), atol=1e-12)
    np.testing.assert_allclose(o.transmat_, xi / xi.sum(1, keepdims=True), atol=1e-12)
    mu = (gamma.T @ X) / gamma.sum(0)[:, None]
    np.testing.assert_allclose(o.means_, mu, atol=1e-12)
    for k in range(K):
        d = X - mu[k]
        cv = (1e-2 + (gamma[:, k, None, None] * (d[:, :, None] * d[:, None, :])).sum(0)) / gamma[:, k].sum()
        np.testing.assert_allclose(o.covars_[k], cv, atol=1e-10)
