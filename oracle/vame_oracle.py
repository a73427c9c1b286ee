"""CPU oracle for the VAME RNN-VAE hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain numpy (float32) restatement of the arithmetic the reference delegates to
PyTorch on its train/embed path.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module; ``vame_amd`` never does.

Parity status: the reference ships no tests or golden vectors, so by the reference's
own material this path is unpinned.  It is pinned instead by fixtures generated in the
build container by importing the reference modules themselves
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``); ``tests/test_oracle.py``
checks every function below against those fixtures.

Every function cites the reference lines (``/root/reference/...``) it restates.
The GRU cell equations are those of ``torch.nn.GRU`` (gate row order r, z, n), which is
what ``vame/model/rnn_model.py:34,91,125`` instantiate.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

F32 = np.float32


def _f32(a):
    return np.ascontiguousarray(a, dtype=F32)


def sigmoid(x):
    return (F32(1.0) / (F32(1.0) + np.exp(-x))).astype(F32)


def softplus(x):
    # torch.nn.Softplus(beta=1, threshold=20): rnn_model.py:61
    return np.where(x > 20, x, np.log1p(np.exp(np.minimum(x, 20)))).astype(F32)


# ----------------------------------------------------------------------------- GRU
def gru_dir_forward(x, h0, W_ih, W_hh, b_ih, b_hh, reverse=False):
    """One direction of one nn.GRU layer, batch_first.  x (B,T,I) -> out (B,T,H), hN.

    Cell (torch.nn.GRU semantics, used at rnn_model.py:34-35, 91-92, 125-126):
      gi = x W_ih^T + b_ih ; gh = h W_hh^T + b_hh
      r = s(gi_r+gh_r) ; u = s(gi_z+gh_z) ; n = tanh(gi_n + r*gh_n) ; h' = (1-u) n + u h
    """
    x = _f32(x)
    B, T, _ = x.shape
    H = W_hh.shape[1]
    h = np.zeros((B, H), F32) if h0 is None else _f32(h0)
    out = np.zeros((B, T, H), F32)
    cache = []
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        gi = x[:, t] @ W_ih.T + b_ih
        gh = h @ W_hh.T + b_hh
        r = sigmoid(gi[:, :H] + gh[:, :H])
        u = sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        ghn = gh[:, 2 * H:]
        n = np.tanh(gi[:, 2 * H:] + r * ghn).astype(F32)
        hn = ((F32(1) - u) * n + u * h).astype(F32)
        cache.append((t, r, u, n, ghn, h))
        out[:, t] = hn
        h = hn
    return out, h, cache


def gru_dir_backward(x, cache, dout, dhN, W_ih, W_hh):
    """BPTT for gru_dir_forward.  Returns dx, dh0, dW_ih, dW_hh, db_ih, db_hh."""
    B, T, I = x.shape
    H = W_hh.shape[1]
    dx = np.zeros((B, T, I), F32)
    dW_ih = np.zeros_like(W_ih)
    dW_hh = np.zeros_like(W_hh)
    db_ih = np.zeros(3 * H, F32)
    db_hh = np.zeros(3 * H, F32)
    dh = np.zeros((B, H), F32) if dhN is None else _f32(dhN).copy()
    for (t, r, u, n, ghn, hprev) in reversed(cache):
        if dout is not None:
            dh = dh + dout[:, t]
        dn = dh * (F32(1) - u)
        du = dh * (hprev - n)
        dhp = dh * u
        dan = dn * (F32(1) - n * n)
        dr = dan * ghn
        dau = du * u * (F32(1) - u)
        dar = dr * r * (F32(1) - r)
        dgi = np.concatenate([dar, dau, dan], 1).astype(F32)
        dgh = np.concatenate([dar, dau, dan * r], 1).astype(F32)
        dx[:, t] = dgi @ W_ih
        dW_ih += dgi.T @ x[:, t]
        dW_hh += dgh.T @ hprev
        db_ih += dgi.sum(0)
        db_hh += dgh.sum(0)
        dh = (dhp + dgh @ W_hh).astype(F32)
    return dx, dh, dW_ih, dW_hh, db_ih, db_hh


def _gru_params(p, prefix, layer, reverse):
    sfx = f"_l{layer}" + ("_reverse" if reverse else "")
    return (p[f"{prefix}.weight_ih{sfx}"], p[f"{prefix}.weight_hh{sfx}"],
            p[f"{prefix}.bias_ih{sfx}"], p[f"{prefix}.bias_hh{sfx}"])


# ----------------------------------------------------------------------------- model
@dataclass
class Spec:
    """Model hyper-parameters (the RNN_VAE ctor args, rnn_model.py:148-150)."""
    T: int = 30          # encoder length = TEMPORAL_WINDOW/2  (rnn_model.py:154)
    F: int = 24
    Z: int = 30
    H: int = 256
    FS: int = 15
    future: bool = True
    softplus: bool = False
    dropout: float = 0.0   # dropout_encoder: nn.GRU(dropout=p) between the two encoder layers, training only (rnn_model.py:34-35)


@dataclass
class FwdCache:
    items: dict = field(default_factory=dict)


def encoder_forward(p, x, cache=None, drop_mask=None, dropout=0.0):
    """rnn_model.py:40-45 : 2-layer bi-GRU, returns cat(h_n[0..3]) = [l0f|l0b|l1f|l1b].
    drop_mask (B,T,2H) in {0,1} with dropout = p: torch.nn.GRU's inter-layer dropout in training (rnn_model.py:34-35) -- layer 1
    reads y0 * mask / (1-p); the final states of layer 0 are returned undropped."""
    pre = "encoder.encoder_rnn"
    o0f, h0f, c0f = gru_dir_forward(x, None, *_gru_params(p, pre, 0, False), reverse=False)
    o0b, h0b, c0b = gru_dir_forward(x, None, *_gru_params(p, pre, 0, True), reverse=True)
    y0 = np.concatenate([o0f, o0b], 2)
    dscale = None
    if drop_mask is not None and dropout > 0:
        dscale = (_f32(drop_mask) / F32(1.0 - dropout)).astype(F32)
        y0 = (y0 * dscale).astype(F32)
    o1f, h1f, c1f = gru_dir_forward(y0, None, *_gru_params(p, pre, 1, False), reverse=False)
    o1b, h1b, c1b = gru_dir_forward(y0, None, *_gru_params(p, pre, 1, True), reverse=True)
    if cache is not None:
        cache.items.update(x=x, y0=y0, c0f=c0f, c0b=c0b, c1f=c1f, c1b=c1b, dscale=dscale)
    return np.concatenate([h0f, h0b, h1f, h1b], 1)


def lambda_forward(p, h, eps, spec, training, cache=None):
    """rnn_model.py:63-76."""
    mu = h @ p["lmbda.hidden_to_mean.weight"].T + p["lmbda.hidden_to_mean.bias"]
    lv_raw = h @ p["lmbda.hidden_to_logvar.weight"].T + p["lmbda.hidden_to_logvar.bias"]
    lv = softplus(lv_raw) if spec.softplus else lv_raw
    mu, lv = _f32(mu), _f32(lv)
    if training:
        std = np.exp(F32(0.5) * lv).astype(F32)
        z = (eps * std + mu).astype(F32)
    else:
        std = None
        z = mu
    if cache is not None:
        cache.items.update(h_n=h, mu=mu, logvar=lv, lv_raw=_f32(lv_raw), std=std, eps=eps, z=z)
    return z, mu, lv


def decoder_forward(p, z, steps, name, rnn, cache=None, inputs=None):
    """rnn_model.py:99-109 / 133-144.

    hidden = Linear(z) (B,2H) ; hidden.view(2,B,H) is a raw reinterpretation of the
    contiguous buffer (rnn_model.py:104,137) -- reproduced by reshape on a C-contiguous array.
    The GRU input is z at every step (rnn_model.py:169-170, 139) unless the caller hands the module its own `inputs` (B, >=steps, Z)
    (the modules run the GRU over whatever they are given, rnn_model.py:106 / :139 `inputs[:, :future_steps, :]`).
    """
    B = z.shape[0]
    hid = _f32(z @ p[f"{name}.latent_to_hidden.weight"].T + p[f"{name}.latent_to_hidden.bias"])
    H = hid.shape[1] // 2
    h0 = hid.reshape(2, B, H)
    ins = np.repeat(z[:, None, :], steps, 1) if inputs is None else _f32(inputs[:, :steps, :])
    pre = f"{name}.{rnn}"
    of, _, cf = gru_dir_forward(ins, h0[0], *_gru_params(p, pre, 0, False), reverse=False)
    ob, _, cb = gru_dir_forward(ins, h0[1], *_gru_params(p, pre, 0, True), reverse=True)
    y = np.concatenate([of, ob], 2)
    pred = _f32(y @ p[f"{name}.hidden_to_output.weight"].T + p[f"{name}.hidden_to_output.bias"])
    if cache is not None:
        cache.items[name] = dict(ins=ins, y=y, cf=cf, cb=cb, hid=hid)
    return pred


def model_forward(p, x, eps, spec, training=True, cache=None, drop_mask=None):
    """RNN_VAE.forward, rnn_model.py:162-179.  x (B,T,F) f32."""
    h_n = encoder_forward(p, x, cache, drop_mask if training else None, spec.dropout)
    z, mu, lv = lambda_forward(p, h_n, eps, spec, training, cache)
    pred = decoder_forward(p, z, spec.T, "decoder", "rnn_rec", cache)
    fut = decoder_forward(p, z, spec.FS, "decoder_future", "rnn_pred", cache) if spec.future else None
    return pred, fut, z, mu, lv


# ----------------------------------------------------------------------------- losses
def mse_loss(x, x_tilde, reduction="sum"):
    """rnn_vae.py:35-43 (nn.MSELoss(reduction))."""
    d = (x_tilde - x).astype(F32)
    s = np.sum(d.astype(np.float64) ** 2)
    return F32(s if reduction == "sum" else s / d.size)


def kl_loss(mu, logvar):
    """rnn_vae.py:53-60 : -0.5 * mean(1 + logvar - mu^2 - exp(logvar))."""
    v = 1 + logvar.astype(np.float64) - mu.astype(np.float64) ** 2 - np.exp(logvar.astype(np.float64))
    return F32(-0.5 * v.mean())


def cluster_loss_svd(latent, kloss, lmbda, batch_size):
    """rnn_vae.py:45-50 as written: (B,B) Gram of latent.T, SVD, lambda*sum(sqrt(sv[:k]))."""
    Hm = latent.T.astype(F32)
    gram = (Hm.T @ Hm) / F32(batch_size)
    sv2 = np.linalg.svd(gram.astype(np.float64), compute_uv=False)
    return F32(lmbda * np.sqrt(np.maximum(sv2[:kloss], 0)).sum())


def cluster_loss_gram(latent, kloss, lmbda, batch_size):
    """Same quantity from the (Z,Z) Gram (what the HIP path computes).

    eig(latent^T latent / B) are the non-zero eig of the (B,B) matrix above.
    Returns (loss, dlatent) with dlatent = lmbda/B * latent V_k S_k^-1 V_k^T.
    """
    z = latent.astype(np.float64)
    G = z.T @ z / batch_size
    w, V = np.linalg.eigh(G)
    # the (B,B) Gram has min(B,Z) non-zero singular values; sv_2[:kloss] can take no more than B of them
    idx = np.argsort(w)[::-1][:min(kloss, z.shape[0], z.shape[1])]
    s = np.sqrt(np.maximum(w[idx], 0))
    Vk = V[:, idx]
    loss = lmbda * s.sum()
    M = (Vk / np.maximum(s, 1e-30)) @ Vk.T
    dz = lmbda / batch_size * z @ M
    return F32(loss), dz.astype(F32)


def kl_annealing(epoch, kl_start, annealtime, function):
    """rnn_vae.py:63-81."""
    if epoch > kl_start:
        if function == "linear":
            return min(1, (epoch - kl_start) / annealtime)
        if function == "sigmoid":
            return float(1 / (1 + np.exp(-0.9 * (epoch - annealtime))))
        raise NotImplementedError('currently only "linear" and "sigmoid" are implemented')
    return 0


def total_loss(pred, fut, z, mu, lv, x, xfut, spec, kl_weight, beta=1.0, kloss=None, klmbda=0.1,
               bsize=None, mse_red="sum", mse_pred="sum"):
    """rnn_vae.py:124-129 (train) ; returns dict of terms."""
    kloss = spec.Z if kloss is None else kloss
    bsize = z.shape[0] if bsize is None else bsize
    rec = mse_loss(x, pred, mse_red)
    futl = mse_loss(xfut, fut, mse_pred) if fut is not None else F32(0)
    kl = kl_loss(mu, lv)
    km, _ = cluster_loss_gram(z, kloss, klmbda, bsize)
    tot = F32(rec + futl + beta * kl_weight * kl + kl_weight * km)
    return dict(rec=rec, fut=futl, kl=kl, kmeans=km, total=tot)


# ----------------------------------------------------------------------------- backward
def model_backward(p, cache, spec, x, xfut, kl_weight, beta=1.0, kloss=None, klmbda=0.1,
                   bsize=None, mse_red="sum", mse_pred="sum"):
    """Gradient of total_loss wrt every parameter (manual BPTT); returns dict name->grad."""
    c = cache.items
    kloss = spec.Z if kloss is None else kloss
    B = x.shape[0]
    bsize = B if bsize is None else bsize
    g = {k: np.zeros_like(v) for k, v in p.items()}
    z, mu, lv, eps, std = c["z"], c["mu"], c["logvar"], c["eps"], c["std"]
    dz = np.zeros_like(z)

    def dec_back(name, rnn, target, red):
        d = c[name]
        y, ins = d["y"], d["ins"]
        Wo = p[f"{name}.hidden_to_output.weight"]
        pred = y @ Wo.T + p[f"{name}.hidden_to_output.bias"]
        scale = F32(2.0) if red == "sum" else F32(2.0 / pred.size)
        dpred = (scale * (pred - target)).astype(F32)
        Bq, Tq, _ = dpred.shape
        g[f"{name}.hidden_to_output.weight"] += dpred.reshape(Bq * Tq, -1).T @ y.reshape(Bq * Tq, -1)
        g[f"{name}.hidden_to_output.bias"] += dpred.sum((0, 1))
        dy = (dpred @ Wo).astype(F32)
        H = y.shape[2] // 2
        pre = f"{name}.{rnn}"
        dzl = np.zeros_like(z)
        dh0 = []
        for rev, cc, sl in ((False, d["cf"], slice(0, H)), (True, d["cb"], slice(H, 2 * H))):
            W_ih, W_hh, _, _ = _gru_params(p, pre, 0, rev)
            dx, dh, dWi, dWh, dbi, dbh = gru_dir_backward(ins, cc, np.ascontiguousarray(dy[:, :, sl]), None, W_ih, W_hh)
            sfx = "_l0" + ("_reverse" if rev else "")
            g[f"{pre}.weight_ih{sfx}"] += dWi
            g[f"{pre}.weight_hh{sfx}"] += dWh
            g[f"{pre}.bias_ih{sfx}"] += dbi
            g[f"{pre}.bias_hh{sfx}"] += dbh
            dzl += dx.sum(1)
            dh0.append(dh)
        dhid = np.stack(dh0, 0).reshape(B, 2 * H)          # inverse of the .view(2,B,H)
        Wl = p[f"{name}.latent_to_hidden.weight"]
        g[f"{name}.latent_to_hidden.weight"] += dhid.T @ z
        g[f"{name}.latent_to_hidden.bias"] += dhid.sum(0)
        dzl += dhid @ Wl
        return dzl

    dz += dec_back("decoder", "rnn_rec", x, mse_red)
    if spec.future:
        dz += dec_back("decoder_future", "rnn_pred", xfut, mse_pred)
    if kl_weight != 0:
        _, dzk = cluster_loss_gram(z, kloss, klmbda, bsize)
        dz += F32(kl_weight) * dzk
    # reparameterisation (rnn_model.py:71-74) + KL (rnn_vae.py:59)
    nBZ = mu.size
    dmu = dz + F32(beta * kl_weight / nBZ) * mu
    dlv = dz * eps * F32(0.5) * std + F32(beta * kl_weight * 0.5 / nBZ) * (np.exp(lv) - F32(1))
    if spec.softplus:
        dlv = dlv * sigmoid(c["lv_raw"])
    dmu, dlv = _f32(dmu), _f32(dlv)
    h_n = c["h_n"]
    g["lmbda.hidden_to_mean.weight"] += dmu.T @ h_n
    g["lmbda.hidden_to_mean.bias"] += dmu.sum(0)
    g["lmbda.hidden_to_logvar.weight"] += dlv.T @ h_n
    g["lmbda.hidden_to_logvar.bias"] += dlv.sum(0)
    dh_n = _f32(dmu @ p["lmbda.hidden_to_mean.weight"] + dlv @ p["lmbda.hidden_to_logvar.weight"])
    H = spec.H
    pre = "encoder.encoder_rnn"
    y0 = c["y0"]
    dy0 = np.zeros_like(y0)
    for rev, cc, off in ((False, c["c1f"], 2 * H), (True, c["c1b"], 3 * H)):
        W_ih, W_hh, _, _ = _gru_params(p, pre, 1, rev)
        dx, _, dWi, dWh, dbi, dbh = gru_dir_backward(y0, cc, None, dh_n[:, off:off + H], W_ih, W_hh)
        sfx = "_l1" + ("_reverse" if rev else "")
        g[f"{pre}.weight_ih{sfx}"] += dWi
        g[f"{pre}.weight_hh{sfx}"] += dWh
        g[f"{pre}.bias_ih{sfx}"] += dbi
        g[f"{pre}.bias_hh{sfx}"] += dbh
        dy0 += dx
    if c.get("dscale") is not None:
        dy0 = (dy0 * c["dscale"]).astype(F32)
    for rev, cc, off in ((False, c["c0f"], 0), (True, c["c0b"], H)):
        W_ih, W_hh, _, _ = _gru_params(p, pre, 0, rev)
        sl = slice(0, H) if not rev else slice(H, 2 * H)
        _, _, dWi, dWh, dbi, dbh = gru_dir_backward(c["x"], cc, np.ascontiguousarray(dy0[:, :, sl]),
                                                    dh_n[:, off:off + H], W_ih, W_hh)
        sfx = "_l0" + ("_reverse" if rev else "")
        g[f"{pre}.weight_ih{sfx}"] += dWi
        g[f"{pre}.weight_hh{sfx}"] += dWh
        g[f"{pre}.bias_ih{sfx}"] += dbi
        g[f"{pre}.bias_hh{sfx}"] += dbh
    return g


# ----------------------------------------------------------------------------- optimiser
def adam_amsgrad_step(p, g, state, lr, step, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam(amsgrad=True) single step (rnn_vae.py:332,143); state = (m, v, vmax) dicts."""
    m, v, vmax = state
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    for k in p:
        m[k] = (b1 * m[k] + (1 - b1) * g[k]).astype(F32)
        v[k] = (b2 * v[k] + (1 - b2) * g[k] * g[k]).astype(F32)
        vmax[k] = np.maximum(vmax[k], v[k])
        denom = (np.sqrt(vmax[k]) / math.sqrt(bc2) + eps).astype(F32)
        p[k] = (p[k] - (lr / bc1) * m[k] / denom).astype(F32)


# ----------------------------------------------------------------------------- batcher / embedding
def window_gather(Xn, starts, length):
    """dataloader.py:45-56 + rnn_vae.py:108 : out[b,t,f] = Xn[f, start_b + t]  -> (B,length,F)."""
    idx = np.asarray(starts)[:, None] + np.arange(length)[None, :]
    return np.ascontiguousarray(np.transpose(Xn[:, idx], (1, 2, 0)))


def normalise_series(X, mean, std):
    """dataloader.py:54 : (sequence - mean)/std in float64 (cast to f32 happens at rnn_vae.py:111-115)."""
    return (X - mean) / std


def embed_series(p, data, spec, batch=256):
    """pose_segmentation.py:84-98 : windows i in [0, N-T), eval-mode mu, float32 (N-T, Z)."""
    N = data.shape[1]
    n = N - spec.T
    out = np.zeros((n, spec.Z), F32)
    for s in range(0, n, batch):
        e = min(n, s + batch)
        x = window_gather(data, np.arange(s, e), spec.T).astype(F32)
        h = encoder_forward(p, x)
        _, mu, _ = lambda_forward(p, h, None, spec, training=False)
        out[s:e] = mu
    return out
