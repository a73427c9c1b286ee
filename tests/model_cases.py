"""Shared model-level parity checks against the golden vectors produced by the reference (tests/golden)."""
import numpy as np
import torch

from conftest import golden_weights, load_golden
from tolerances import TINY_REL, assert_grad_close, assert_loss_close, step_scale_of
from vame_amd import ops
from vame_amd.model.rnn_model import RNN_VAE


def build_model(g, dev):
    T, F, Z, H, FS, fut, sp = [int(v) for v in g["spec"][:7]]
    model = RNN_VAE(2 * T, Z, F, fut, FS, H, H, H, H, 0, 0, 0, bool(sp))
    w = golden_weights(g)
    if w:
        model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    return model.to(dev), (T, F, Z, H, FS, fut, sp)


def check_step(dev, name, kw, mse="sum", tol_grad=TINY_REL, via_autograd=False, stepwise=False):
    g = load_golden(name)
    model, (T, F, Z, H, FS, fut, sp) = build_model(g, dev)
    model.train()
    if stepwise:                                  # force the large-H per-step path on a small model
        model._ensure_engine().stepwise = True
    x, xfut, eps = [torch.from_numpy(g[k]).to(dev) for k in ("x", "xfut", "eps")]
    B = x.shape[0]
    tag = f"kw{kw:g}/"
    ref = g[tag + "losses"]
    if not via_autograd:
        win = torch.cat([x, xfut], 1).contiguous()
        out = model.loss_step(win, kw, beta=1.0, kloss=Z, klmbda=0.1, bsize=B, mse_red=mse, mse_pred=mse, eps=eps).cpu().numpy()
        for i, k in enumerate(["rec", "fut", "kl", "kmeans"]):
            assert_loss_close(out[i], ref[i], name=k)
    else:
        from vame_amd.model import rnn_vae as rv
        res = model(x, eps=eps)
        pred, z, mu, lv = res[0], res[-3], res[-2], res[-1]
        loss = rv.reconstruction_loss(x, pred, mse) + kw * rv.kullback_leibler_loss(mu, lv) + kw * rv.cluster_loss(z.T, Z, 0.1, B)
        if fut:
            loss = loss + rv.future_reconstruction_loss(xfut, res[1], mse)
        for p in model.parameters():
            p.grad = None
        loss.backward()
        assert_loss_close(loss.item(), ref[4], name="total")
    if "pred" in g and (tag + "losses") == max(k for k in g if k.endswith("/losses")):
        eng = model._engine
        np.testing.assert_allclose(eng.buf("mu", B, Z)[:B * Z].view(B, Z).cpu().numpy(), g["mu"], atol=1e-5)
        np.testing.assert_allclose(eng.buf("pred", B, T, F)[:B * T * F].view(B, T, F).cpu().numpy(), g["pred"], atol=3e-5)
    sscale = step_scale_of(g[tag + "g/" + k] for k, _ in model.named_parameters())
    for k, p in model.named_parameters():
        assert_grad_close(p.grad.cpu().numpy(), g[tag + "g/" + k], tol_grad, k, sscale)      # relative to each tensor's own max
    return model


def check_eval_and_submodules(dev):
    g = load_golden("step_tiny")
    model, (T, F, Z, H, FS, fut, sp) = build_model(g, dev)
    model.eval()
    x = torch.from_numpy(g["x"]).to(dev)
    pred, futp, z, mu, lv = model(x)
    np.testing.assert_allclose(mu.cpu().numpy(), g["eval_mu"], atol=1e-5)
    np.testing.assert_allclose(pred.cpu().numpy(), g["eval_pred"], atol=3e-5)
    assert z is mu
    # the call pattern of pose_segmentation.py:92-95 and generative_functions.py:39
    h = model.encoder(x)
    m2, _, _ = model.lmbda(h)
    np.testing.assert_allclose(m2.cpu().numpy(), g["eval_mu"], atol=1e-5)
    ins = mu.unsqueeze(2).repeat(1, 1, T).permute(0, 2, 1)
    np.testing.assert_allclose(model.decoder(ins, mu).cpu().numpy(), g["eval_pred"], atol=3e-5)
    sd = model.state_dict()
    assert list(sd.keys()) == list(golden_weights(g).keys())
    assert all(tuple(sd[k].shape) == golden_weights(g)[k].shape for k in sd)


def check_evaluate_and_generative_cores(dev):
    """Numeric cores of vame.evaluate_model / vame.generative_model against the reference's eval-mode
    golden outputs: reconstruct_test_batch = model(data) in eval, decode_latents = model.decoder(tiled z, z)."""
    from vame_amd.analysis.generative_functions import decode_latents
    from vame_amd.model.evaluate import reconstruct_test_batch
    g = load_golden("step_tiny")
    model, (T, F, Z, H, FS, fut, sp) = build_model(g, dev)
    model.eval()
    win = torch.from_numpy(np.concatenate([g["x"], g["xfut"]], 1)).to(dev)
    r = reconstruct_test_batch(model, win, T, fut, FS)
    np.testing.assert_array_equal(r["data"], g["x"])
    np.testing.assert_array_equal(r["fut_orig"], g["xfut"])
    np.testing.assert_allclose(r["data_tilde"], g["eval_pred"], atol=3e-5)
    np.testing.assert_allclose(r["mu"], g["eval_mu"], atol=1e-5)
    assert r["fut"].shape == g["xfut"].shape
    np.testing.assert_allclose(decode_latents(model, g["eval_mu"], T), g["eval_pred"], atol=3e-5)


def check_h0_view(dev):
    g = load_golden("h0view")
    T, F, Z, H = [int(v) for v in g["spec"]]
    model = RNN_VAE(2 * T, Z, F, 0, 0, H, H, H, H, 0, 0, 0, False)
    sd = model.state_dict()
    for k, v in golden_weights(g).items():
        sd[k] = torch.from_numpy(v)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    for B in (1, 2, 6):
        z = torch.from_numpy(g[f"B{B}/z"]).to(dev)
        pred = model.decoder(None, z)
        np.testing.assert_allclose(pred.cpu().numpy(), g[f"B{B}/pred"], atol=2e-5)


def check_decoder_inputs(dev):
    """model.decoder(inputs, z) / model.decoder_future(inputs, z) with inputs that are NOT z tiled over time (the reference's
    modules run their GRU over any sequence, rnn_model.py:99-109,132-144): against the reference's own outputs."""
    g = load_golden("decoder_inputs")
    T, F, Z, H, FS = [int(v) for v in g["spec"]]
    model = RNN_VAE(2 * T, Z, F, 1, FS, H, H, H, H, 0, 0, 0, False)
    sd = model.state_dict()
    for k, v in golden_weights(g).items():
        sd[k] = torch.from_numpy(v)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    for B in (1, 5):
        z, ins = torch.from_numpy(g[f"B{B}/z"]).to(dev), torch.from_numpy(g[f"B{B}/ins"]).to(dev)
        np.testing.assert_allclose(model.decoder(ins, z).cpu().numpy(), g[f"B{B}/pred"], atol=2e-5)
        np.testing.assert_allclose(model.decoder_future(ins, z).cpu().numpy(), g[f"B{B}/fut"], atol=2e-5)
        tiled = z.unsqueeze(2).repeat(1, 1, T).permute(0, 2, 1)                 # the usual call still takes the z-only path
        assert not np.allclose(model.decoder(tiled, z).cpu().numpy(), g[f"B{B}/pred"], atol=1e-3)
    import pytest
    with pytest.raises(ValueError):
        model.decoder(torch.zeros(1, T + 2, Z), torch.zeros(1, Z))


def check_padded_hidden_sizes(dev):
    """Hidden sizes that are not multiples of 32 (torch.nn.GRU accepts any): the reference's own step at H = 40 (step_h40.npz, all
    gradients), the sub-module call pattern and state_dict shapes, then H = 100 with different decoder sizes and encoder dropout
    against the numpy oracle."""
    from oracle import vame_oracle as vo
    model = check_step(dev, "step_h40", 1.0)
    check_step(dev, "step_h40", 1.0, via_autograd=True)
    g = load_golden("step_h40")
    assert all(tuple(v.shape) == golden_weights(g)[k].shape for k, v in model.state_dict().items())
    model.eval()
    x = torch.from_numpy(g["x"]).to(dev)
    h = model.encoder(x)
    assert tuple(h.shape) == (x.shape[0], 4 * 40)
    np.testing.assert_allclose(model.lmbda(h)[1].cpu().numpy(), g["eval_mu"], atol=1e-5)
    np.testing.assert_allclose(model(x)[0].cpu().numpy(), g["eval_pred"], atol=3e-5)
    # H = 100 encoder, 72 / 24 decoders, inter-layer dropout with an injected mask
    T, F, Z, FS, B = 7, 12, 10, 3, 9
    torch.manual_seed(23)
    model = RNN_VAE(2 * T, Z, F, 1, FS, 100, 100, 72, 24, 0.2, 0, 0, False)
    p = {k: v.numpy().copy() for k, v in model.state_dict().items()}
    model = model.to(dev).train()
    rng = np.random.default_rng(12)
    win = rng.standard_normal((B, T + FS, F)).astype(np.float32)
    eps = rng.standard_normal((B, Z)).astype(np.float32)
    mask = (rng.random((B, T, 200)) > 0.2).astype(np.float32)
    out = model.loss_step(torch.from_numpy(win).to(dev), 0.6, beta=1.0, kloss=Z, klmbda=0.1, bsize=B, eps=torch.from_numpy(eps).to(dev),
                          drop_mask=torch.from_numpy(mask).to(dev)).cpu().numpy()
    spec = vo.Spec(T=T, F=F, Z=Z, H=100, FS=FS, future=True, softplus=False, dropout=0.2)
    cache = vo.FwdCache()
    x, xf = win[:, :T], win[:, T:]
    res = vo.model_forward(p, x, eps, spec, True, cache, drop_mask=mask)
    L = vo.total_loss(*res, x, xf, spec, 0.6)
    for i, k in enumerate(["rec", "fut", "kl", "kmeans"]):
        assert_loss_close(out[i], L[k], name=k)
    grads = vo.model_backward(p, cache, spec, x, xf, 0.6)
    for k, prm in model.named_parameters():
        assert tuple(prm.grad.shape) == p[k].shape
        assert_grad_close(prm.grad.cpu().numpy(), grads[k], TINY_REL, k, step_scale_of(grads.values()))


def check_legacy_padded_hidden(dev):
    """RNN_VAE_LEGACY with a hidden size that is not a multiple of 32 (the padded parameter image under the legacy parameter names:
    encoder.rnn_1 / rnn_2, lmbda.hidden_to_linear, uni-directional decoder) against the stock-torch restatement of the legacy model
    (oracle/torch_ref.py, itself pinned by step_legacy.npz): loss terms and every gradient."""
    from oracle.torch_ref import TorchRefLegacy, reference_loss
    from vame_amd.model.rnn_model import RNN_VAE_LEGACY
    T, F, Z, H, FS, B = 6, 12, 10, 40, 3, 7
    torch.manual_seed(4)
    m = RNN_VAE_LEGACY(2 * T, Z, F, 1, FS, H, H, H, H, 0, 0, 0, False)
    ref = TorchRefLegacy(T, F, Z, H, FS)
    ref.load_state_dict({k: v.clone() for k, v in m.state_dict().items()})
    ref.train()
    gen = torch.Generator().manual_seed(1)
    win, eps = torch.randn(B, T + FS, F, generator=gen), torch.randn(B, Z, generator=gen)
    loss, terms = reference_loss(ref(win[:, :T], eps), win[:, :T], win[:, T:], 0.7, kloss=Z, bsize=B)
    loss.backward()
    m = m.to(dev).train()
    got = m.loss_step(win.to(dev).contiguous(), 0.7, beta=1.0, kloss=Z, klmbda=0.1, bsize=B, eps=eps.to(dev)).cpu().numpy()
    for i, t in enumerate(terms):
        assert_loss_close(got[i], float(t.detach()), name=str(i))
    rg = {k: p.grad for k, p in ref.named_parameters()}
    sc = step_scale_of(v.numpy() for v in rg.values() if v is not None)
    for k, p in m.named_parameters():
        r = rg[k].numpy() if rg.get(k) is not None else np.zeros(tuple(p.shape), np.float32)     # hidden_to_linear: unused, no gradient
        assert_grad_close(p.grad.cpu().numpy(), r, TINY_REL, k, sc)


def check_noise_input(dev):
    """cfg['noise']: the encoder sees a perturbed input while the reconstruction target stays clean (rnn_vae.py:116-124)."""
    from oracle import vame_oracle as vo
    g = load_golden("step_tiny")
    model, (T, F, Z, H, FS, fut, sp) = build_model(g, dev)
    model.train()
    x, xfut, eps = g["x"], g["xfut"], g["eps"]
    rng = np.random.default_rng(3)
    xin = (x + 0.3 * rng.standard_normal(x.shape)).astype(np.float32)
    win = torch.cat([torch.from_numpy(x), torch.from_numpy(xfut)], 1).contiguous().to(dev)
    out = model.loss_step(win, 1.0, beta=1.0, kloss=Z, klmbda=0.1, bsize=x.shape[0], eps=torch.from_numpy(eps).to(dev),
                          enc_in=torch.from_numpy(xin).to(dev)).cpu().numpy()
    p = golden_weights(g)
    spec = vo.Spec(T=T, F=F, Z=Z, H=H, FS=FS, future=bool(fut), softplus=bool(sp))
    cache = vo.FwdCache()
    res = vo.model_forward(p, xin, eps, spec, True, cache)
    L = vo.total_loss(*res, x, xfut, spec, 1.0)
    for i, k in enumerate(["rec", "fut", "kl", "kmeans"]):
        assert_loss_close(out[i], L[k], name=k)
    grads = vo.model_backward(p, cache, spec, x, xfut, 1.0)
    for k, prm in model.named_parameters():
        r = grads[k]
        assert_grad_close(prm.grad.cpu().numpy(), r, TINY_REL, k, step_scale_of(grads.values()))


def check_adam_trajectory(dev):
    """Three optimizer steps (fused loss_step + FusedAdamAMSGrad) reproduce the reference's torch.optim.Adam(amsgrad=True)
    trajectory recorded in step_tiny.npz (same eps per step): losses per step and all final weights."""
    from vame_amd.model.rnn_vae import FusedAdamAMSGrad
    g = load_golden("step_tiny")
    model, (T, F, Z, H, FS, fut, sp) = build_model(g, dev)
    model.train()
    opt = FusedAdamAMSGrad(model, lr=5e-4)
    win = torch.cat([torch.from_numpy(g["x"]), torch.from_numpy(g["xfut"])], 1).contiguous().to(dev)
    B = win.shape[0]
    for s_ in range(g["adam/eps"].shape[0]):
        terms = model.loss_step(win, 1.0, beta=1.0, kloss=Z, klmbda=0.1, bsize=B, eps=torch.from_numpy(g["adam/eps"][s_]).to(dev)).cpu().numpy()
        total = terms[0] + terms[1] + terms[2] + terms[3]
        assert abs(total - g["adam/loss"][s_]) <= 1e-4 * abs(g["adam/loss"][s_])
        opt.step()
    sd = model.state_dict()
    for k, v in golden_weights(g, "adam/w/").items():
        np.testing.assert_allclose(sd[k].cpu().numpy(), v, atol=2e-5, err_msg=k)


def check_optimizer_checkpoint(dev):
    """FusedAdamAMSGrad.state_dict() / load_state_dict() (torch's Optimizer API, what checkpointing code calls): a run interrupted after two
    steps and resumed in a NEW model + optimizer ends on the golden three-step trajectory's weights, like the uninterrupted run."""
    import io
    from vame_amd.model.rnn_vae import FusedAdamAMSGrad
    g = load_golden("step_tiny")
    win = torch.cat([torch.from_numpy(g["x"]), torch.from_numpy(g["xfut"])], 1).contiguous().to(dev)
    B, n = win.shape[0], g["adam/eps"].shape[0]

    def run(model, opt, steps):
        for s_ in steps:
            model.loss_step(win, 1.0, beta=1.0, kloss=model.spec.Z, klmbda=0.1, bsize=B, eps=torch.from_numpy(g["adam/eps"][s_]).to(dev))
            opt.step()
    model, _ = build_model(g, dev)
    model.train()
    opt = FusedAdamAMSGrad(model, lr=5e-4)
    run(model, opt, range(n - 1))
    blob = io.BytesIO()
    torch.save(dict(model=model.state_dict(), opt=opt.state_dict()), blob)       # (what a user's checkpoint code does)
    blob.seek(0)
    ck = torch.load(blob, map_location=dev, weights_only=False)
    assert ck["opt"]["fused"]["t"] == n - 1 and ck["opt"]["param_groups"][0]["lr"] == 5e-4
    model2, _ = build_model(g, dev)
    model2.train()
    model2.load_state_dict(ck["model"])
    opt2 = FusedAdamAMSGrad(model2, lr=1.0)                 # (a wrong rate on purpose: load_state_dict brings the saved one back)
    opt2.load_state_dict(ck["opt"])
    run(model2, opt2, range(n - 1, n))
    sd = model2.state_dict()
    for k, v in golden_weights(g, "adam/w/").items():
        np.testing.assert_allclose(sd[k].cpu().numpy(), v, atol=2e-5, err_msg=k)


def check_failed_step_leaves_no_sums(dev):
    """A step that raises between the loss kernels and vame_loss_finish_f32 must not leak its partial sums into the next step's terms."""
    import pytest
    g = load_golden("step_tiny")
    model, (T, F, Z, H, FS, fut, sp) = build_model(g, dev)
    model.train()
    win = torch.cat([torch.from_numpy(g["x"]), torch.from_numpy(g["xfut"])], 1).contiguous().to(dev)
    B = win.shape[0]
    eps = torch.from_numpy(g["adam/eps"][0]).to(dev)
    kw = dict(beta=1.0, kloss=Z, klmbda=0.1, bsize=B, eps=eps)
    good = model.loss_step(win, 1.0, **kw).cpu().numpy()
    eng = model._engine
    real = eng.backward

    def boom(*a, **k):
        raise RuntimeError("injected")
    eng.backward = boom
    with pytest.raises(RuntimeError, match="injected"):
        model.loss_step(win, 1.0, **kw)
    eng.backward = real
    assert float(eng.loss_sums().abs().sum()) == 0.0
    again = model.loss_step(win, 1.0, **kw).cpu().numpy()
    np.testing.assert_allclose(again, good, rtol=1e-6)            # (equal up to the order of the per-workgroup atomic adds into the loss sums)


def check_engine_option_validation(dev):
    import pytest
    g = load_golden("step_tiny")
    for bad in (dict(coop_cover=3), dict(coop_cover=-1), dict(coop_rounds=0), dict(small_streams=-1), dict(wgrad_streams=-2), dict(no_such=1)):
        model, _ = build_model(g, dev)
        model.engine_options = bad
        with pytest.raises(ValueError, match=next(iter(bad))):
            model._ensure_engine()


def check_odd_dims_vs_oracle(dev, F=10, Z=7, H=32, T=9, FS=4, B=5, expect=None, engine_options=None):
    """Unaligned shapes (F, Z not multiples of 4; egocentric_data=False gives F = num_features - 2): exercises the scalar-load
    GEMM paths and the non-fused layer-0 input projection, against the numpy oracle (no golden needed)."""
    from oracle import vame_oracle as vo
    torch.manual_seed(3)
    model = RNN_VAE(2 * T, Z, F, 1, FS, H, H, H, H, 0, 0, 0, False)
    p = {k: v.numpy().copy() for k, v in model.state_dict().items()}
    model = model.to(dev).train()
    model.engine_options = dict(engine_options or {})
    rng = np.random.default_rng(9)
    win = rng.standard_normal((B, T + FS, F)).astype(np.float32)
    eps = rng.standard_normal((B, Z)).astype(np.float32)
    out = model.loss_step(torch.from_numpy(win).to(dev), 0.7, beta=2.0, kloss=4, klmbda=0.3, bsize=B, eps=torch.from_numpy(eps).to(dev)).cpu().numpy()
    if expect is not None:
        expect(model._engine)
    spec = vo.Spec(T=T, F=F, Z=Z, H=H, FS=FS)
    cache = vo.FwdCache()
    x, xf = win[:, :T], win[:, T:]
    res = vo.model_forward(p, x, eps, spec, True, cache)
    L = vo.total_loss(*res, x, xf, spec, 0.7, beta=2.0, kloss=4, klmbda=0.3)
    for i, k in enumerate(["rec", "fut", "kl", "kmeans"]):
        assert_loss_close(out[i], L[k], name=k)
    grads = vo.model_backward(p, cache, spec, x, xf, 0.7, beta=2.0, kloss=4, klmbda=0.3)
    for k, prm in model.named_parameters():
        r = grads[k]
        assert_grad_close(prm.grad.cpu().numpy(), r, TINY_REL, k, step_scale_of(grads.values()))


def check_legacy_step(dev):
    """RNN_VAE_LEGACY (cfg['legacy'], reference rnn_model.py:186-324) against one train step of the reference
    (tests/golden/step_legacy.npz): outputs, the four loss terms, all gradients, eval mode, sub-module call patterns."""
    from vame_amd.model.rnn_model import RNN_VAE_LEGACY
    g = load_golden("step_legacy")
    T, F, Z, H, FS, fut, sp, B = [int(v) for v in g["spec"]]
    model = RNN_VAE_LEGACY(2 * T, Z, F, 1, FS, H, H, H, H, 0, 0, 0, False)
    w = {k[2:]: g[k] for k in g if k.startswith("w/")}
    assert list(model.state_dict().keys()) == list(w.keys())
    model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    model = model.to(dev).train()
    x, xfut, eps = [torch.from_numpy(g[k]).to(dev) for k in ("x", "xfut", "eps")]
    win = torch.cat([x, xfut], 1).contiguous()
    kw = float(g["kw"][0])
    out = model.loss_step(win, kw, beta=1.0, kloss=Z, klmbda=0.1, bsize=B, eps=eps).cpu().numpy()
    ref = g["losses"]
    for i in range(4):
        assert_loss_close(out[i], ref[i], name=str(i))
    eng = model._engine
    for name, ref_v in (("pred", g["pred"]), ("futp", g["fut"]), ("z", g["z"]), ("mu", g["mu"]), ("logvar", g["logvar"])):
        got = eng.ws.t[name][:ref_v.size].view(*ref_v.shape).cpu().numpy()
        np.testing.assert_allclose(got, ref_v, atol=3e-5, err_msg=name)
    for k, p in model.named_parameters():
        gr = g["g/" + k]
        assert_grad_close(p.grad.cpu().numpy(), gr, TINY_REL, k, step_scale_of(g["g/" + kk] for kk, _ in model.named_parameters()))
    assert set(g["no_grad"]) == {"lmbda.hidden_to_linear.weight", "lmbda.hidden_to_linear.bias"}
    # autograd path through model(x)
    model.zero_grad(set_to_none=False)
    pred, futp, z, mu, lv = model(x, eps=eps)
    np.testing.assert_allclose(pred.detach().cpu().numpy(), g["pred"], atol=3e-5)
    (pred.sum() + futp.sum()).backward()
    assert float(model.decoder.hidden_to_output.weight.grad.abs().sum()) > 0
    # eval mode + the legacy sub-module signatures: decoder(inputs) takes the tiled latent only (rnn_model.py:263,288)
    model.eval()
    ep, ef, ez, emu, elv = model(x)
    np.testing.assert_allclose(ep.cpu().numpy(), g["eval_pred"], atol=3e-5)
    np.testing.assert_allclose(ef.cpu().numpy(), g["eval_fut"], atol=3e-5)
    np.testing.assert_allclose(emu.cpu().numpy(), g["eval_mu"], atol=1e-5)
    ins = emu.unsqueeze(2).repeat(1, 1, T).permute(0, 2, 1)
    np.testing.assert_allclose(model.decoder(ins).cpu().numpy(), g["eval_pred"], atol=3e-5)
    np.testing.assert_allclose(model.decoder_future(ins).cpu().numpy(), g["eval_fut"], atol=3e-5)
    h = model.encoder(x)
    np.testing.assert_allclose(model.lmbda(h)[1].cpu().numpy(), g["eval_mu"], atol=1e-5)


def check_model_options(dev, name):
    """Reference options beyond the stock config (rnn_model.py:31-35 encoder inter-layer dropout; :148-160 decoder hidden sizes that
    differ from the encoder's): one train step against the reference's own outputs, losses and all gradients
    (tests/golden/step_tiny_dropout.npz, step_tiny_hsizes.npz), plus eval mode (dropout off)."""
    g = load_golden(name)
    T, F, Z, H, FS, fut, sp, B, h2, hrec, hpred = [int(v) for v in g["spec"]]
    pdrop = float(g["dropout"][0])
    model = RNN_VAE(2 * T, Z, F, fut, FS, H, h2, hrec, hpred, pdrop, 0, 0, False)
    w = golden_weights(g)
    assert list(model.state_dict().keys()) == list(w.keys())
    assert all(tuple(model.state_dict()[k].shape) == w[k].shape for k in w)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    model = model.to(dev).train()
    x, xfut, eps = [torch.from_numpy(g[k]).to(dev) for k in ("x", "xfut", "eps")]
    mask = torch.from_numpy(g["drop_mask"]).to(dev) if "drop_mask" in g else None
    win = torch.cat([x, xfut], 1).contiguous()
    kw = float(g["kw"][0])
    out = model.loss_step(win, kw, beta=1.0, kloss=Z, klmbda=0.1, bsize=B, eps=eps, drop_mask=mask).cpu().numpy()
    for i in range(4):
        assert_loss_close(out[i], g["losses"][i], name=str(i))
    eng = model._engine
    for nm, ref_v in (("pred", g["pred"]), ("futp", g["fut"]), ("z", g["z"]), ("mu", g["mu"]), ("logvar", g["logvar"])):
        got = eng.ws.t[nm][:ref_v.size].view(*ref_v.shape).cpu().numpy()
        np.testing.assert_allclose(got, ref_v, atol=3e-5, err_msg=nm)
    for k, p in model.named_parameters():
        gr = g["g/" + k]
        assert_grad_close(p.grad.cpu().numpy(), gr, TINY_REL, k, step_scale_of(g["g/" + kk] for kk, _ in model.named_parameters()))
    # autograd path with the same injected draws
    model.zero_grad(set_to_none=False)
    res = model(x, eps=eps, drop_mask=mask)
    np.testing.assert_allclose(res[0].detach().cpu().numpy(), g["pred"], atol=3e-5)
    (res[0].sum() + res[1].sum()).backward()
    if pdrop > 0:                                   # device-drawn mask: right keep rate, and training still runs
        m = model._dropout_mask(64, None, torch.device(dev))
        assert tuple(m.shape) == (64, T, 2 * H) and abs(float(m.mean()) - (1 - pdrop)) < 0.02 and set(m.unique().tolist()) <= {0.0, 1.0}
        assert np.isfinite(model.loss_step(win, kw, beta=1.0, kloss=Z, klmbda=0.1, bsize=B).cpu().numpy()).all()
    model.eval()
    ep, ef, ez, emu, elv = model(x)
    np.testing.assert_allclose(ep.cpu().numpy(), g["eval_pred"], atol=3e-5)
    np.testing.assert_allclose(ef.cpu().numpy(), g["eval_fut"], atol=3e-5)
    np.testing.assert_allclose(emu.cpu().numpy(), g["eval_mu"], atol=1e-5)


def check_fused_heads_match(dev):
    """engine option fuse_heads (default on: one pass per decoder over its states for output Linear + MSE + dY + the Linear's weight gradient,
    vame_head_stream_f32) gives the losses and gradients of the separate launches (two contractions, the MSE kernel, the split-K weight gradient)."""
    rng = np.random.default_rng(31)
    T, F, Z, H, FS, B = 7, 12, 6, 32, 3, 37
    outs = []
    for fuse in (False, True):
        torch.manual_seed(5)
        model = RNN_VAE(2 * T, Z, F, 1, FS, H, H, H, H, 0, 0, 0, False).to(dev).train()
        eng = model._ensure_engine()
        eng.fuse_heads = fuse
        win = torch.from_numpy(np.random.default_rng(2).standard_normal((B, T + FS, F)).astype(np.float32)).to(dev)
        eps = torch.from_numpy(np.random.default_rng(3).standard_normal((B, Z)).astype(np.float32)).to(dev)
        calls, orig = [], ops.head_stream
        ops.head_stream = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        try:
            terms = model.loss_step(win, 0.8, beta=1.5, kloss=Z, klmbda=0.2, bsize=B, eps=eps).cpu().numpy()
        finally:
            ops.head_stream = orig
        assert len(calls) == (2 if fuse else 0)
        outs.append((terms, model.flat_parameters()[1].clone().cpu().numpy(), eng.buf("pred", B, T, F)[:B * T * F].cpu().numpy()))
    (t0, g0, p0), (t1, g1, p1) = outs
    np.testing.assert_allclose(t1, t0, rtol=2e-5)
    np.testing.assert_allclose(p1, p0, atol=2e-5)
    np.testing.assert_allclose(g1, g0, atol=3e-5 * float(np.abs(g0).max()))


def check_stale_backward_guard(dev):
    """model(x) in training keeps one step's activations in the engine workspace: backward() after another forward must raise
    instead of returning another batch's gradients."""
    import pytest
    g = load_golden("step_tiny")
    model, (T, F, Z, H, FS, fut, sp) = build_model(g, dev)
    model.train()
    x = torch.from_numpy(g["x"]).to(dev)
    out1 = model(x)
    model(x)                                       # overwrites the stashes of the first forward
    with pytest.raises(RuntimeError, match="overwritten"):
        out1[0].sum().backward()
    out2 = model(x)
    out2[0].sum().backward()                       # the latest forward is fine


def check_device_window_loader(dev, tmp_path):
    """DeviceWindowLoader (the product batcher) against the reference's SEQUENCE_DATASET + DataLoader collate
    (tests/golden/batcher.npz, dataloader.py:18-56): same window starts from the global numpy stream, and bit-identical
    windows -- the series is z-scored once in float64 and cast, which equals the reference's per-window normalise-then-cast."""
    import os
    from vame_amd.model.dataloader import SEQUENCE_DATASET, DeviceWindowLoader
    g = load_golden("batcher")
    T2, B = int(g["T2"]), g["batch"].shape[0]
    path = str(tmp_path) + os.sep
    np.save(path + "train_seq.npy", g["X"])
    ds = SEQUENCE_DATASET(path, data="train_seq.npy", train=True, temporal_window=T2)
    assert float(ds.mean) == float(g["mean"]) and float(ds.std) == float(g["std"])
    assert float(np.load(path + "seq_mean.npy")) == float(g["mean"])            # written for the test set / later runs
    loader = DeviceWindowLoader(ds, B, T2, torch.device(dev))
    assert len(loader) == g["X"].shape[1] // B
    np.random.seed(11)
    starts = loader.draw_starts()
    np.testing.assert_array_equal(starts, g["starts"])
    win = loader.gather(starts).cpu().numpy()                                    # (B, 2T, F) fp32
    ref = np.transpose(g["batch"], (0, 2, 1)).astype(np.float32)                 # reference item (B,F,2T) f64 -> permute -> FloatTensor
    np.testing.assert_array_equal(win, ref)
    # the API-compatible __getitem__ (index ignored, float64 (F, 2T) item)
    np.random.seed(11)
    item = ds[123]
    np.testing.assert_array_equal(item.numpy(), g["batch"][0])
    # two ranks: disjoint slices of one draw
    l0, l1 = DeviceWindowLoader(ds, B // 2, T2, torch.device(dev), 0, 2), DeviceWindowLoader(ds, B // 2, T2, torch.device(dev), 1, 2)
    np.random.seed(11)
    s0 = l0.draw_starts()
    np.random.seed(11)
    s1 = l1.draw_starts()
    np.testing.assert_array_equal(np.concatenate([s0, s1]), g["starts"])


def check_coop_failure_is_contained(dev):
    """A cooperative (column-split) GRU launch that reports a hand-off timeout must not reach the weights and must raise
    promptly: the optimizer kernel drops the step on the device (abort word), the host raises at the next step's enqueue
    (asynchronous status snapshot) or at the next synchronising check -- not an epoch later.  Fault injection:
    vame_gru_coop_set_poll_limit(-1)."""
    import pytest
    from vame_amd import _lib, ops
    from vame_amd.model.rnn_vae import FusedAdamAMSGrad
    F, Z, H, T, FS, B = 10, 7, 128, 4, 2, 5
    torch.manual_seed(3)
    model = RNN_VAE(2 * T, Z, F, 1, FS, H, H, H, H, 0, 0, 0, False).to(dev).train()
    opt = FusedAdamAMSGrad(model, lr=5e-4)
    rng = np.random.default_rng(9)
    win = torch.from_numpy(rng.standard_normal((B, T + FS, F)).astype(np.float32)).to(dev)
    model.loss_step(win, 1.0, beta=1.0, kloss=4, klmbda=0.3, bsize=B)
    opt.step()
    eng = model._engine
    assert eng._coop_state is not None and eng._coop_state.dirty          # this shape runs the cooperative kernels
    eng.check_async_errors()                                                # clean so far
    w0 = model.flat_parameters()[0].clone()
    old = ops.gru_coop_set_poll_limit(-1)
    raised = False
    try:
        try:
            model.loss_step(win, 1.0, beta=1.0, kloss=4, klmbda=0.3, bsize=B)          # every cooperative launch reports a timeout
            opt.step()                                                                  # ... so this is dropped on the device
            ops.gru_coop_set_poll_limit(0)
            model.loss_step(win, 1.0, beta=1.0, kloss=4, klmbda=0.3, bsize=B)          # raises here, when the snapshot has arrived,
            model.loss_step(win, 1.0, beta=1.0, kloss=4, klmbda=0.3, bsize=B)          # or at the latest one step later
            eng.check_async_errors()
        except _lib.VameHipError as e:
            raised = "hand-off" in str(e)
    finally:
        ops.gru_coop_set_poll_limit(old if old > 0 else 0)
    assert raised
    if dev != "cpu":
        torch.cuda.synchronize()
    assert torch.equal(model.flat_parameters()[0], w0)                                  # the failed step never reached the weights
    # after the exception the status word is clean again and training continues
    model.loss_step(win, 1.0, beta=1.0, kloss=4, klmbda=0.3, bsize=B)
    opt.step()
    eng.check_async_errors()
    assert not torch.equal(model.flat_parameters()[0], w0)
    # inference entry points check too: a failed launch never hands out results
    ops.gru_coop_set_poll_limit(-1)
    try:
        model.eval()
        with pytest.raises(_lib.VameHipError, match="hand-off"):
            model(win[:, :T].contiguous())
    finally:
        ops.gru_coop_set_poll_limit(0)
    model(win[:, :T].contiguous())
