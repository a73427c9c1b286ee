"""Reference-equivalent PyTorch-CPU model -- TEST INFRASTRUCTURE / CPU BASELINE ONLY.

A restatement of the reference's train step built from stock torch modules (nn.GRU, F.linear,
the (B,B) torch.svd cluster loss, Adam-AMSGrad) so the CPU baseline of bench.py times the same
arithmetic the reference runs (vame/model/rnn_model.py:162-179, vame/model/rnn_vae.py:35-60,124-143)
without any reference file travelling to the GPU box.  Checked against the golden vectors in
tests/test_oracle.py.  Never imported by vame_amd.
"""
import torch
import torch.nn.functional as Fn
from torch import nn


class TorchRef(nn.Module):
    def __init__(self, T=30, F=24, Z=30, H=256, FS=15, future=True, softplus=False):
        super().__init__()
        self.T, self.F, self.Z, self.H, self.FS, self.future, self.softplus = T, F, Z, H, FS, future, softplus
        # registration order = the reference's (Encoder GRU, 2 Lambda Linears, Decoder GRU/l2h/h2o, Future ...)
        self.rnn = nn.ModuleDict()
        self.lin = nn.ModuleDict()
        self.rnn["enc"] = nn.GRU(F, H, num_layers=2, batch_first=True, bidirectional=True)
        self.lin["mean"] = nn.Linear(4 * H, Z)
        self.lin["logvar"] = nn.Linear(4 * H, Z)
        for tag in ("dec", "fut") if future else ("dec",):
            self.rnn[tag] = nn.GRU(Z, H, num_layers=1, batch_first=True, bidirectional=True)
            self.lin[tag + "_l2h"] = nn.Linear(Z, 2 * H)
            self.lin[tag + "_out"] = nn.Linear(2 * H, F)

    KEYMAP = {"rnn.enc": "encoder.encoder_rnn", "lin.mean": "lmbda.hidden_to_mean", "lin.logvar": "lmbda.hidden_to_logvar",
              "rnn.dec": "decoder.rnn_rec", "lin.dec_l2h": "decoder.latent_to_hidden", "lin.dec_out": "decoder.hidden_to_output",
              "rnn.fut": "decoder_future.rnn_pred", "lin.fut_l2h": "decoder_future.latent_to_hidden",
              "lin.fut_out": "decoder_future.hidden_to_output"}

    def load_reference_state(self, sd):
        """sd uses the reference's state_dict keys."""
        own = {}
        for k in self.state_dict():
            mod, leaf = k.rsplit(".", 1)
            own[k] = torch.as_tensor(sd[self.KEYMAP[mod] + "." + leaf])
        self.load_state_dict(own)

    def reference_named_grads(self):
        return {self.KEYMAP[k.rsplit(".", 1)[0]] + "." + k.rsplit(".", 1)[1]: p.grad for k, p in self.named_parameters()}

    def _decode(self, tag, z, steps):
        B = z.shape[0]
        h0 = self.lin[tag + "_l2h"](z).view(2, B, self.H)            # raw reinterpretation, as the reference does
        y, _ = self.rnn[tag](z.unsqueeze(1).expand(B, steps, self.Z).contiguous(), h0)
        return self.lin[tag + "_out"](y)

    def forward(self, x, eps=None):
        _, hn = self.rnn["enc"](x)
        h = torch.cat([hn[0], hn[1], hn[2], hn[3]], 1)
        mu = self.lin["mean"](h)
        lv = self.lin["logvar"](h)
        if self.softplus:
            lv = Fn.softplus(lv)
        if self.training:
            if eps is None:
                eps = torch.randn_like(mu)
            z = eps * torch.exp(0.5 * lv) + mu
        else:
            z = mu
        pred = self._decode("dec", z, self.T)
        fut = self._decode("fut", z, self.FS) if self.future else None
        return pred, fut, z, mu, lv


class TorchRefLegacy(nn.Module):
    """Stock-torch restatement of RNN_VAE_LEGACY (reference rnn_model.py:186-324): two stacked 1-layer bi-GRUs, an unused
    hidden_to_linear layer, softplus on the log-variance always, uni-directional reconstruction decoder, zero initial states.
    Sub-module names are the reference's, so its state_dict loads directly.  Pinned by tests/golden/step_legacy.npz."""

    def __init__(self, T=30, F=24, Z=30, H=256, FS=15, future=True):
        super().__init__()
        self.T, self.F, self.Z, self.H, self.FS, self.future = T, F, Z, H, FS, future
        self.encoder = nn.ModuleDict(dict(rnn_1=nn.GRU(F, H, batch_first=True, bidirectional=True),
                                          rnn_2=nn.GRU(2 * H, H, batch_first=True, bidirectional=True)))
        self.lmbda = nn.ModuleDict(dict(hidden_to_linear=nn.Linear(4 * H, 4 * H), hidden_to_mean=nn.Linear(4 * H, Z),
                                        hidden_to_logvar=nn.Linear(4 * H, Z)))
        self.decoder = nn.ModuleDict(dict(rnn_rec=nn.GRU(Z, H, batch_first=True, bidirectional=False),
                                          hidden_to_output=nn.Linear(H, F)))
        if future:
            self.decoder_future = nn.ModuleDict(dict(rnn_pred=nn.GRU(Z, H, batch_first=True, bidirectional=True),
                                                     hidden_to_output=nn.Linear(2 * H, F)))

    def forward(self, x, eps=None):
        o1, h1 = self.encoder["rnn_1"](x)
        _, h2 = self.encoder["rnn_2"](o1)
        h = torch.cat([h1[0], h1[1], h2[0], h2[1]], 1)
        mu = self.lmbda["hidden_to_mean"](h)
        lv = Fn.softplus(self.lmbda["hidden_to_logvar"](h))
        if self.training:
            if eps is None:
                eps = torch.randn_like(mu)
            z = eps * torch.exp(0.5 * lv) + mu
        else:
            z = mu
        ins = z.unsqueeze(1).expand(z.shape[0], self.T, self.Z).contiguous()
        pred = self.decoder["hidden_to_output"](self.decoder["rnn_rec"](ins)[0])
        fut = None
        if self.future:
            fut = self.decoder_future["hidden_to_output"](self.decoder_future["rnn_pred"](ins[:, :self.FS])[0])
        return pred, fut, z, mu, lv


def reference_loss(out, x, xfut, kl_weight, beta=1.0, kloss=30, klmbda=0.1, bsize=None, red="sum"):
    pred, fut, z, mu, lv = out
    bsize = z.shape[0] if bsize is None else bsize
    rec = Fn.mse_loss(pred, x, reduction=red)
    fl = Fn.mse_loss(fut, xfut, reduction=red) if fut is not None else torch.zeros(())
    gram = (z @ z.T) / bsize                                           # (B,B): H.T @ H with H = latent.T
    _, sv2, _ = torch.svd(gram)
    km = klmbda * torch.sqrt(sv2[:kloss]).sum()
    kl = -0.5 * torch.mean(1 + lv - mu.pow(2) - lv.exp())
    return rec + fl + beta * kl_weight * kl + kl_weight * km, (rec, fl, kl, km)


def time_train_steps(B=256, steps=10, warmup=3, threads=None, seed=19, budget_s=25.0):
    """CPU baseline: windows/s of fwd + loss (incl. the (B,B) SVD) + bwd + Adam-AMSGrad at the default config.
    Bounded: stops after `steps` timed steps or `budget_s` seconds of timed work, whichever comes first."""
    import os
    import time
    threads = threads or len(os.sched_getaffinity(0))
    torch.set_num_threads(threads)
    torch.manual_seed(seed)
    m = TorchRef()
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=5e-4, amsgrad=True)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, 30, 24, generator=g)
    xf = torch.randn(B, 15, 24, generator=g)
    t0, done = None, 0
    for i in range(warmup + steps):
        if i == warmup:
            t0 = time.perf_counter()
        loss, _ = reference_loss(m(x), x, xf, 1.0)
        opt.zero_grad()
        loss.backward()
        opt.step()
        if t0 is not None:
            done += 1
            if time.perf_counter() - t0 > budget_s:
                break
    dt = time.perf_counter() - t0
    steps = done
    return dict(value=B * steps / dt, unit="windows/s", cores=threads, kind="port",
                sample=f"{steps} train steps of B={B} (T=30,F=24,H=256,Z=30,FS=15, fp32, torch {torch.__version__} CPU nn.GRU + (B,B) svd "
                       f"+ Adam-amsgrad), {dt:.1f} s")


def time_embed(batch=1, budget_s=10.0, threads=None, seed=19, n_frames=200_000):
    """CPU baseline of the embedding leg: the reference's loop (vame/analysis/pose_segmentation.py:87-98) -- window i of the (F,N)
    series -> (batch,T,F) float32 -> encoder -> mean head, eval mode -- with `batch` windows per forward (1 = as written).
    Bounded: stops after `budget_s` seconds of timed work."""
    import os
    import time
    import numpy as np
    threads = threads or len(os.sched_getaffinity(0))
    torch.set_num_threads(threads)
    torch.manual_seed(seed)
    m = TorchRef()
    m.eval()
    T = m.T
    rng = np.random.default_rng(0)
    data = rng.standard_normal((m.F, n_frames)).astype(np.float32)
    done, t0 = 0, None
    with torch.no_grad():
        i = 0
        for it in range(10 ** 9):
            if it == 2:
                t0, done = time.perf_counter(), 0
            x = torch.from_numpy(np.stack([data[:, j:j + T].T for j in range(i, i + batch)]))
            _, hn = m.rnn["enc"](x)
            mu = m.lin["mean"](torch.cat([hn[0], hn[1], hn[2], hn[3]], 1))
            mu.numpy()
            i = (i + batch) % (n_frames - T - batch)
            done += batch
            if t0 is not None and time.perf_counter() - t0 > budget_s:
                break
    dt = time.perf_counter() - t0
    return dict(value=done / dt, unit="windows/s", cores=threads, kind="port",
                sample=f"{done} windows in {dt:.1f} s, batch {batch}, torch {torch.__version__} CPU nn.GRU")
