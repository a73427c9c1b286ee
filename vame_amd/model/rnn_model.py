"""RNN-VAE model classes with the reference's names, constructor signatures and state_dict layout,
whose forward passes run on the MI355X HIP kernels (vame_amd.engine) instead of torch.nn.GRU.

Drop-in surface kept (reference: vame/model/rnn_model.py):
  Encoder :23-45, Lambda :48-76, Decoder :79-109, Decoder_Future :112-144, RNN_VAE :147-179,
  the *_LEGACY variants :186-324 (cfg['legacy'])
  * the same sub-module / parameter names, so `state_dict()` keys, shapes and default
    initialisation under `torch.manual_seed` are identical and .pkl checkpoints interchange;
  * `model.encoder(x)`, `model.lmbda(h)`, `model.decoder(ins, z)` stay callable on their own
    (pose_segmentation.py:92-95, generative_functions.py:39).
The torch.nn.GRU / nn.Linear objects are parameter containers only; they are never executed.
"""
import torch
from torch import nn

from .. import _lib, ops
from ..engine import ParamTable, Spec, VAEEngine
from ..padding import PadMap, needs_padding


class _EngineOwner:
    """Mixin giving sub-modules access to the RNN_VAE that owns the flat buffers + engine."""
    _owner = None

    def _eng(self):
        if self._owner is None:
            raise RuntimeError(f"{type(self).__name__} must be used as part of an RNN_VAE (it runs on the shared HIP engine)")
        return self._owner[0]._ensure_engine()


class Encoder(nn.Module, _EngineOwner):
    def __init__(self, NUM_FEATURES, hidden_size_layer_1, hidden_size_layer_2, dropout_encoder):
        super().__init__()
        self.input_size, self.hidden_size, self.hidden_size_2 = NUM_FEATURES, hidden_size_layer_1, hidden_size_layer_2
        self.n_layers, self.dropout, self.bidirectional = 2, dropout_encoder, True
        self.encoder_rnn = nn.GRU(input_size=NUM_FEATURES, hidden_size=hidden_size_layer_1, num_layers=2, bias=True,
                                  batch_first=True, dropout=dropout_encoder, bidirectional=True)
        self.hidden_factor = 4

    def forward(self, inputs):
        """(B,T,F) -> (B,4H) = [l0 fwd | l0 bwd | l1 fwd | l1 bwd] final states (rnn_model.py:40-45)."""
        eng = self._eng()
        x = _as_f32(inputs, eng.dev)
        B, T, F = x.shape
        _check(T == eng.spec.T and F == eng.spec.F, f"encoder input {tuple(x.shape)} != (B,{eng.spec.T},{eng.spec.F})")
        with torch.no_grad():
            hn = eng.encode(x, T * F, B, training=False)
            out = hn[:B * 4 * eng.spec.H].view(B, 4 * eng.spec.H).clone()
            pad = self._owner[0]._pad
            if pad is not None:                      # hidden size not a multiple of 32: drop the padded units (vame_amd/padding.py)
                out = pad.unpad_cols(out, 4, pad.true_spec.H)
        eng.check_async_errors()
        return out


class Lambda(nn.Module, _EngineOwner):
    def __init__(self, ZDIMS, hidden_size_layer_1, hidden_size_layer_2, softplus):
        super().__init__()
        self.hid_dim, self.latent_length, self.softplus = hidden_size_layer_1 * 4, ZDIMS, softplus
        self.hidden_to_mean = nn.Linear(self.hid_dim, ZDIMS)
        self.hidden_to_logvar = nn.Linear(self.hid_dim, ZDIMS)
        if softplus == True:  # noqa: E712  (config values may be 0/1)
            print("Using a softplus activation to ensures that the variance is parameterized as non-negative and "
                  "activated by a smooth function")
            self.softplus_fn = nn.Softplus()

    def forward(self, hidden, eps=None):
        """(B,4H) -> (z, mean, logvar); eval mode returns (mean, mean, logvar) (rnn_model.py:63-76)."""
        eng = self._eng()
        h = _as_f32(hidden, eng.dev)
        pad = self._owner[0]._pad
        if pad is not None:
            h = pad.pad_cols(h, 4, pad.true_spec.H)
        B, Z = h.shape[0], eng.spec.Z
        with torch.no_grad():
            if eps is not None:
                eps = _as_f32(eps, eng.dev)
            z, mu, lv = eng.latent(h, B, eps, self.training, want_kl=False)      # eps = None in training mode: drawn by the kernel
            self.mean, self.logvar = mu[:B * Z].view(B, Z).clone(), lv[:B * Z].view(B, Z).clone()
            zz = z[:B * Z].view(B, Z).clone() if self.training else self.mean
        return zz, self.mean, self.logvar


class _DecoderBase(nn.Module, _EngineOwner):
    def _run(self, inputs, z, which):
        eng = self._eng()
        zt = _as_f32(z, eng.dev)
        B = zt.shape[0]
        # `inputs` is z tiled over time in every reference caller (rnn_model.py:169-170, generative_functions.py:36-39): then the
        # kernels read z once.  Any other sequence (the modules run their GRU over whatever they are given, rnn_model.py:106,139-140)
        # takes the per-step input projection the encoder's second layer uses.
        seq = None
        if inputs is not None and torch.is_tensor(inputs):
            steps = eng.spec.T if which == "dec" else eng.spec.FS
            # Decoder returns one output per input step (rnn_model.py:106-109) and the model is built for seq_len steps;
            # Decoder_Future reads inputs[:, :future_steps] (rnn_model.py:139)
            ok_len = inputs.dim() == 3 and (inputs.shape[1] == steps if which == "dec" else inputs.shape[1] >= steps)
            _check(ok_len and inputs.shape[0] == B and inputs.shape[2] == eng.spec.Z,
                   f"decoder inputs {tuple(inputs.shape)}: expected (B, {'' if which == 'dec' else '>='}{steps}, {eng.spec.Z})")
            it = _as_f32(inputs, eng.dev)
            tiled = bool(torch.equal(it[:, :steps, :], zt[:, None, :].expand(B, steps, eng.spec.Z)))
            if not tiled:
                seq = (it, it.shape[1])
        with torch.no_grad():
            pred, fut = eng.decode(zt, B, training=False, which=which, inputs=seq)
            s = eng.spec
            out = pred[:B * s.T * s.F].view(B, s.T, s.F).clone() if which == "dec" else fut[:B * s.FS * s.F].view(B, s.FS, s.F).clone()
        eng.check_async_errors()
        return out


class Decoder(_DecoderBase):
    def __init__(self, TEMPORAL_WINDOW, ZDIMS, NUM_FEATURES, hidden_size_rec, dropout_rec):
        super().__init__()
        self.num_features, self.sequence_length, self.hidden_size = NUM_FEATURES, TEMPORAL_WINDOW, hidden_size_rec
        self.latent_length, self.n_layers, self.dropout, self.bidirectional = ZDIMS, 1, dropout_rec, True
        self.rnn_rec = nn.GRU(ZDIMS, hidden_size=hidden_size_rec, num_layers=1, bias=True, batch_first=True,
                              dropout=dropout_rec, bidirectional=True)
        self.hidden_factor = 2
        self.latent_to_hidden = nn.Linear(ZDIMS, hidden_size_rec * 2)
        self.hidden_to_output = nn.Linear(hidden_size_rec * 2, NUM_FEATURES)

    def forward(self, inputs, z):
        return self._run(inputs, z, "dec")


class Decoder_Future(_DecoderBase):
    def __init__(self, TEMPORAL_WINDOW, ZDIMS, NUM_FEATURES, FUTURE_STEPS, hidden_size_pred, dropout_pred):
        super().__init__()
        self.num_features, self.future_steps, self.sequence_length = NUM_FEATURES, FUTURE_STEPS, TEMPORAL_WINDOW
        self.hidden_size, self.latent_length, self.n_layers = hidden_size_pred, ZDIMS, 1
        self.dropout, self.bidirectional = dropout_pred, True
        self.rnn_pred = nn.GRU(ZDIMS, hidden_size=hidden_size_pred, num_layers=1, bias=True, batch_first=True,
                               dropout=dropout_pred, bidirectional=True)
        self.hidden_factor = 2
        self.latent_to_hidden = nn.Linear(ZDIMS, hidden_size_pred * 2)
        self.hidden_to_output = nn.Linear(hidden_size_pred * 2, NUM_FEATURES)

    def forward(self, inputs, z):
        return self._run(inputs, z, "fut")


def _as_f32(t, dev):
    if not torch.is_tensor(t):
        t = torch.as_tensor(t)
    return t.detach().to(device=dev, dtype=torch.float32).contiguous()


def _check(cond, msg):
    if not cond:
        raise ValueError(msg)


class _VAEFunction(torch.autograd.Function):
    """Autograd bridge for `model(x)` in training mode: HIP forward, HIP backward into the flat bucket."""

    @staticmethod
    def forward(ctx, model, x, eps, drop_mask, *params):
        eng = model._ensure_engine()
        B = x.shape[0]
        outs = eng.forward(x, x.shape[1] * x.shape[2], B, eps, training=True, drop_mask=drop_mask)
        ctx.model, ctx.B, ctx.x, ctx.serial = model, B, x, eng.serial
        return tuple(o.clone() if o is not None else None for o in outs)

    @staticmethod
    def backward(ctx, dpred, dfut, dz, dmu, dlv):
        model, B = ctx.model, ctx.B
        eng = model._ensure_engine()
        s = eng.spec
        if eng.serial != ctx.serial:
            # the BPTT stashes / sequences of this forward live in the engine's shared workspace (one step's worth, ~11 GB at
            # batch 4096): another forward of the same model has overwritten them, so these gradients would be another batch's
            raise RuntimeError("vame_amd.RNN_VAE: backward() of a forward pass whose activations were overwritten by a later "
                               "model(x) / loss_step / encoder / decoder call; call backward() before the next forward "
                               "(one outstanding training forward per model)")

        def seed(name, g, n):
            buf = eng.buf(name, n)
            if g is None:
                buf[:n].zero_()
            else:
                buf[:n].copy_(g.contiguous().view(-1))
        seed("dpred", dpred, B * s.T * s.F)
        if s.future:
            seed("dfut", dfut, B * s.FS * s.F)
        c = lambda g: None if g is None else g.contiguous()
        if model._pad is not None:                   # the engine writes the padded gradient image; the real entries go to the temp bucket
            eng.backward(B, 0.0, 0.0, dz_ext=c(dz), dmu_ext=c(dmu), dlv_ext=c(dlv), use_minv=False)
            model._pad.pull_grads(model._flat_gtmp)
        else:
            eng.g = model._flat_gtmp
            try:
                eng.backward(B, 0.0, 0.0, dz_ext=c(dz), dmu_ext=c(dmu), dlv_ext=c(dlv), use_minv=False)
            finally:
                eng.g = model._flat_g
        model._accumulate_tmp_grads()
        return (None, None, None, None) + (None,) * len(model._param_list)


class RNN_VAE(nn.Module):
    _LEGACY = False

    def __init__(self, TEMPORAL_WINDOW, ZDIMS, NUM_FEATURES, FUTURE_DECODER, FUTURE_STEPS, hidden_size_layer_1,
                 hidden_size_layer_2, hidden_size_rec, hidden_size_pred, dropout_encoder, dropout_rec, dropout_pred, softplus):
        super().__init__()
        _check(not self._LEGACY or (hidden_size_layer_2 == hidden_size_layer_1 == hidden_size_rec
                                    and (not FUTURE_DECODER or hidden_size_pred == hidden_size_layer_1)),
               "vame_amd: RNN_VAE_LEGACY on the gfx950 kernels needs one hidden size for all GRUs")
        _check(0 <= float(dropout_encoder) < 1, f"dropout_encoder={dropout_encoder} must be in [0, 1)")
        # dropout_rec / dropout_pred: torch.nn.GRU ignores dropout for num_layers=1 (it only warns), so they have no effect in
        # the reference either (rnn_model.py:91-92,125-126); dropout_encoder acts between the two encoder layers in training
        self.FUTURE_DECODER = FUTURE_DECODER
        self.seq_len = int(TEMPORAL_WINDOW / 2)
        self._build_modules(ZDIMS, NUM_FEATURES, FUTURE_DECODER, FUTURE_STEPS, hidden_size_layer_1, hidden_size_layer_2,
                            hidden_size_rec, hidden_size_pred, dropout_encoder, dropout_rec, dropout_pred, softplus)
        self.spec = Spec(T=self.seq_len, F=NUM_FEATURES, Z=ZDIMS, H=hidden_size_layer_1, FS=FUTURE_STEPS if FUTURE_DECODER else 0,
                         future=bool(FUTURE_DECODER), softplus=bool(softplus) or self._LEGACY, legacy=self._LEGACY,
                         H_rec=0 if hidden_size_rec == hidden_size_layer_1 else hidden_size_rec,
                         H_pred=0 if (not FUTURE_DECODER or hidden_size_pred == hidden_size_layer_1) else hidden_size_pred,
                         dropout=0.0 if self._LEGACY else float(dropout_encoder))
        for m in (self.encoder, self.lmbda, self.decoder, getattr(self, "decoder_future", None)):
            if m is not None:
                object.__setattr__(m, "_owner", (self,))       # tuple: not registered as a sub-module
        self._flat_p = self._flat_g = self._flat_gtmp = None
        self._engine = None
        self._pad = None
        self._bucket_check = None
        # options of the engine built on first use (vame_amd.engine.ENGINE_DEFAULTS: kernel choice / scheduling; optional, the defaults are
        # the measured best).  train_model() fills it from the optional `vame_amd_engine:` mapping of config.yaml.  Set before the first call.
        self.engine_options = {}
        self._register_state_dict_hook(_clone_state_dict)

    def _build_modules(self, ZDIMS, NUM_FEATURES, FUTURE_DECODER, FUTURE_STEPS, h1, h2, h_rec, h_pred, d_enc, d_rec, d_pred, softplus):
        self.encoder = Encoder(NUM_FEATURES, h1, h2, d_enc)
        self.lmbda = Lambda(ZDIMS, h1, h2, softplus)
        self.decoder = Decoder(self.seq_len, ZDIMS, NUM_FEATURES, h_rec, d_rec)
        if FUTURE_DECODER:
            self.decoder_future = Decoder_Future(self.seq_len, ZDIMS, NUM_FEATURES, FUTURE_STEPS, h_pred, d_pred)

    # ---------------------------------------------------------------- flat parameter bucket
    def _ensure_engine(self, touch=True):
        """The engine over the flat parameter bucket (built on first use / after the model moved).  touch: the caller is about to run the
        model, so the weights may have changed since the last call -> refresh the padded image and have the GRU packs rebuilt;
        bookkeeping callers (flat_parameters: optimizer, all-reduce) pass False and leave both alone."""
        # fast path (every step calls this; walking named_parameters() costs ~0.3 ms of a host-bound 2.5 ms step at the stock batch):
        # the Parameter objects the flat bucket was built over still point into it.  .cuda() / .to() / .float() replace p.data (the
        # addresses change), load_state_dict copies in place (they do not)
        chk = self._bucket_check
        ok = chk is not None
        if ok:
            # the LIVE parameter of every slot is still the object the bucket was built over (module surgery, load_state_dict(assign=True) or
            # `mod.weight = nn.Parameter(...)` replace the object: the old one would go on pointing into the bucket) and still points into it
            for owner, name, p_, addr in chk:
                if owner._parameters.get(name) is not p_ or p_.data_ptr() != addr:
                    ok = False
                    break
        plist = None
        if not ok:
            plist = list(self.named_parameters())
            dev = plist[0][1].device
            _lib.require_device_tensor(plist[0][1])     # "move the model with .cuda()": there is no CPU fallback
            ok = self._flat_p is not None and self._flat_p.device == dev
            if ok:
                base, esz, tab = self._flat_p.data_ptr(), 4, self._table
                ok = all(p.data_ptr() == base + esz * tab.off(n) for n, p in plist)
                if ok:
                    self._bucket_check = self._slots(plist)
        if not ok:
            self._table = ParamTable([(n, p.shape) for n, p in plist])
            flat_p = torch.zeros(self._table.numel, device=dev)
            # four floats behind the gradients: slot 0 is the status word that rides the same all-reduce (rnn_vae.allreduce_gradients)
            self._flat_g_comm = torch.zeros(self._table.numel + 4, device=dev)
            flat_g = self._flat_g_comm[:self._table.numel]
            for n, p in plist:
                o, k = self._table.off(n), p.numel()
                flat_p[o:o + k].copy_(p.data.detach().reshape(-1).to(torch.float32))
                p.data = flat_p[o:o + k].view(p.shape)
                p.grad = flat_g[o:o + k].view(p.shape)
            self._flat_p, self._flat_g = flat_p, flat_g
            self._flat_gtmp = torch.zeros_like(flat_g)
            self._param_list = [p for _, p in plist]
            self._bucket_check = self._slots(plist)
            if needs_padding(self.spec):
                # a hidden size that is not a multiple of 32 (torch.nn.GRU takes any): the kernels run on a zero-padded image of
                # the parameters; the model, its state_dict, the optimizer and the all-reduce keep the reference's shapes
                self._pad = PadMap(self.spec, self._table, [(n, tuple(p.shape)) for n, p in plist], dev)
                self._engine = VAEEngine(self._pad.spec, self._pad.table, self._pad.p, self._pad.g, options=self.engine_options)
            else:
                self._pad = None
                self._engine = VAEEngine(self.spec, self._table, flat_p, flat_g, options=self.engine_options)
        if touch or not ok:
            if self._pad is not None:
                self._pad.push_params(self._flat_p)
            self._engine.version += 1          # weights may have changed since the last call: repack (a few tiny kernels)
        return self._engine

    def _slots(self, plist):
        """[(owning module, attribute name, Parameter, address)] of every parameter: what _ensure_engine's fast path re-checks per call."""
        out = []
        for n, p in plist:
            mod, _, leaf = n.rpartition(".")
            out.append((self.get_submodule(mod) if mod else self, leaf, p, p.data_ptr()))
        return out

    def _accumulate_tmp_grads(self):
        for n, p in self.named_parameters():
            o, k = self._table.off(n), p.numel()
            gview = self._flat_g[o:o + k].view(p.shape)
            tview = self._flat_gtmp[o:o + k].view(p.shape)
            if p.grad is None:
                gview.copy_(tview)
                p.grad = gview
            else:
                p.grad.add_(tview)

    def flat_parameters(self):
        """(flat_p, flat_g): the contiguous fp32 parameter / gradient buckets (RCCL all-reduce + fused Adam)."""
        self._ensure_engine(touch=False)
        return self._flat_p, self._flat_g

    # ---------------------------------------------------------------- forward
    def _dropout_mask(self, B, drop_mask, dev):
        """{0,1} keep-mask (B, T, 2H) of the encoder's inter-layer dropout (training, dropout_encoder > 0), drawn on the device
        unless injected (parity tests)."""
        s = self.spec
        if not (self.training and s.dropout > 0):
            return None
        if drop_mask is None:
            drop_mask = torch.bernoulli(torch.full((B, s.T, 2 * s.H), 1.0 - s.dropout, device=dev))
        drop_mask = drop_mask.to(device=dev, dtype=torch.float32).contiguous()
        _check(tuple(drop_mask.shape) == (B, s.T, 2 * s.H), f"drop_mask {tuple(drop_mask.shape)} != {(B, s.T, 2 * s.H)}")
        return drop_mask

    def _engine_mask(self, mask):
        """The keep-mask in the engine's (possibly padded) hidden-unit layout."""
        if mask is None or self._pad is None:
            return mask
        return self._pad.pad_cols(mask, 2, self.spec.H, fill=1.0).contiguous()

    def forward(self, seq, eps=None, drop_mask=None):
        """rnn_model.py:162-179.  Returns (prediction, future, z, mu, logvar) or, without the
        future decoder, (prediction, z, mu, logvar).  `eps` optionally injects the N(0,1) draw of
        the reparameterisation, `drop_mask` the encoder's dropout keep-mask (parity tests); by default both are drawn on
        the device."""
        eng = self._ensure_engine()
        s = self.spec
        x = seq.to(device=eng.dev, dtype=torch.float32).contiguous()
        _check(x.dim() == 3 and x.shape[1] == s.T and x.shape[2] == s.F, f"input {tuple(x.shape)} != (B,{s.T},{s.F})")
        B = x.shape[0]
        if self.training and eps is not None:            # (None: the latent kernel draws it -- device-side Philox stream, VAEEngine.seed_rng)
            eps = eps.to(device=eng.dev, dtype=torch.float32).contiguous()
        drop_mask = self._engine_mask(self._dropout_mask(B, drop_mask, eng.dev))
        if self.training and torch.is_grad_enabled():
            pred, fut, z, mu, lv = _VAEFunction.apply(self, x, eps, drop_mask, *self._param_list)
        else:
            with torch.no_grad():
                outs = eng.forward(x, s.T * s.F, B, eps, training=self.training, drop_mask=drop_mask)
                pred, fut, z, mu, lv = [o.clone() if o is not None else None for o in outs]
            if not self.training:
                z = mu
            eng.check_async_errors()
        self.lmbda.mean, self.lmbda.logvar = mu, lv
        if self.FUTURE_DECODER:
            return pred, fut, z, mu, lv
        return pred, z, mu, lv

    # ---------------------------------------------------------------- fused training / evaluation step
    def loss_step(self, win, kl_weight, *, beta, kloss, klmbda, bsize, mse_red="sum", mse_pred="sum", eps=None, backward=True,
                  enc_in=None, drop_mask=None, weights=None, acc=None):
        """One fused forward + loss (+ backward) over a batch of windows, entirely on device.

        win: (B, L, F) fp32 device tensor, L >= T (+FS): steps [0,T) are the encoder input and
        reconstruction target, steps [T, T+FS) the future target (rnn_vae.py:111-112).
        `enc_in` (B,T,F) optionally replaces the encoder input (cfg['noise'], rnn_vae.py:116-119); targets stay `win`.
        Gradients land in the flat bucket (p.grad views).  Returns a device tensor
        [rec, fut, kl, kmeans] in the reference's units (no host sync).  `acc` (float64[6] device tensor, optional): the same launch adds
        [total, rec, fut, kl, kmeans] to acc[0:5] and stores this step's total in acc[5], total = sum(weights x terms) with `weights`
        defaulting to the reference's (1, 1, beta kl_weight, kl_weight) (rnn_vae.py:129-150's bookkeeping, on the device)."""
        eng = self._ensure_engine()
        s = self.spec
        B, L, F = win.shape
        training = self.training
        eng.poll_async_errors()
        _check(F == s.F and L >= s.T + (s.FS if (training and s.future) else 0), f"window batch {tuple(win.shape)} too short")
        try:
            return self._loss_step(eng, win, kl_weight, beta, kloss, klmbda, bsize, mse_red, mse_pred, eps, backward, enc_in, drop_mask, weights, acc)
        except BaseException:
            # the loss kernels ADD into the 8-float sum buffer and only vame_loss_finish_f32 -- the last launch of a step -- zeroes it: a step that
            # raised in between (a shape check, a VameHipError, an out-of-memory the caller catches) must not leak its partial sums into the next
            if not eng.capturing:
                try:
                    eng.abandon_step()
                except Exception:
                    pass
            raise

    def _loss_step(self, eng, win, kl_weight, beta, kloss, klmbda, bsize, mse_red, mse_pred, eps, backward, enc_in, drop_mask, weights, acc):
        s = self.spec
        B, L, F = win.shape
        training = self.training
        with torch.no_grad():
            if enc_in is not None:
                enc_in = enc_in.to(device=eng.dev, dtype=torch.float32).contiguous()
                _check(tuple(enc_in.shape) == (B, s.T, F), f"enc_in {tuple(enc_in.shape)} != {(B, s.T, F)}")
            # eps = None in training mode: drawn inside the latent kernel (the reference's torch.randn_like, rnn_model.py:71-74)
            eng.forward(win, L * F, B, eps, training, cluster=(kl_weight, kloss, klmbda, bsize), enc_in=enc_in,
                        drop_mask=self._engine_mask(self._dropout_mask(B, drop_mask, eng.dev)), defer_heads=True, want_kl=True)
            # test(): no future term (rnn_vae.py:183-198)
            losses = eng.loss(B, win, L * F, s.T * F, kl_weight, kloss, klmbda, bsize, mse_red, mse_pred,
                              with_future=training and s.future)
            if backward and training:
                eng.backward(B, kl_weight, beta)
                if self._pad is not None:
                    self._pad.pull_grads(self._flat_g)
            eng.join_cluster()
            # [rec, fut, kl, kmeans] in the reference's units, their weighted total and the caller's epoch accumulators: ONE single-thread
            # launch (vame_loss_finish_f32), which also leaves the sums zeroed for the next step -- no torch op, no vendor BLAS call
            with_fut = bool(training and s.future)
            scale = (1.0 if mse_red == "sum" else 1.0 / (B * s.T * F), 1.0 if (not s.future or mse_pred == "sum") else 1.0 / (B * s.FS * F),
                     -0.5 / (B * s.Z), 1.0)
            out5 = torch.empty(5, device=eng.dev)                  # (allocation only; a fresh tensor per step: callers may keep the terms)
            ops.loss_finish(losses, scale, weights if weights is not None else (1.0, 1.0, beta * kl_weight, kl_weight), with_fut, out5, acc)
            out = out5[:4]
        eng.snapshot_async_errors()
        return out

    def load_state_dict(self, state_dict, *a, **k):
        r = super().load_state_dict(state_dict, *a, **k)
        if self._engine is not None:
            self._engine.version += 1
        return r


# ------------------------------------------------------------------------------------------------ legacy topology
# RNN_VAE_LEGACY (reference rnn_model.py:186-324, selected by cfg['legacy']): the encoder is two stacked 1-layer
# bidirectional GRUs (the same arithmetic as the 2-layer one when both hidden sizes agree), Lambda always applies softplus to
# the log-variance and carries an unused `hidden_to_linear` layer, the reconstruction decoder is UNI-directional, and neither
# decoder gets an initial state from z.  Parameter names / order follow the reference so checkpoints interchange.
class Encoder_LEGACY(Encoder):
    def __init__(self, NUM_FEATURES, hidden_size_layer_1, hidden_size_layer_2, dropout_encoder):
        nn.Module.__init__(self)
        self.input_size, self.hidden_size, self.hidden_size_2 = NUM_FEATURES, hidden_size_layer_1, hidden_size_layer_2
        self.n_layers, self.dropout = 1, dropout_encoder
        self.rnn_1 = nn.GRU(input_size=NUM_FEATURES, hidden_size=hidden_size_layer_1, num_layers=1, bias=True, batch_first=True,
                            dropout=dropout_encoder, bidirectional=True)
        self.rnn_2 = nn.GRU(input_size=hidden_size_layer_1 * 2, hidden_size=hidden_size_layer_2, num_layers=1, bias=True,
                            batch_first=True, dropout=dropout_encoder, bidirectional=True)


class Lambda_LEGACY(Lambda):
    def __init__(self, ZDIMS, hidden_size_layer_1, hidden_size_layer_2):
        nn.Module.__init__(self)
        self.hid_dim, self.latent_length = hidden_size_layer_1 * 2 + hidden_size_layer_2 * 2, ZDIMS
        self.hidden_to_linear = nn.Linear(self.hid_dim, self.hid_dim)        # never used by forward (rnn_model.py:223,229-242)
        self.hidden_to_mean = nn.Linear(self.hid_dim, ZDIMS)
        self.hidden_to_logvar = nn.Linear(self.hid_dim, ZDIMS)
        self.softplus = nn.Softplus()

    def forward(self, cell_output, eps=None):
        z, mean, logvar = Lambda.forward(self, cell_output, eps)
        self.latent_mean, self.latent_logvar = mean, logvar
        return z, mean, logvar


class Decoder_LEGACY(_DecoderBase):
    def __init__(self, TEMPORAL_WINDOW, ZDIMS, NUM_FEATURES, hidden_size_rec, dropout_rec):
        super().__init__()
        self.num_features, self.sequence_length, self.hidden_size = NUM_FEATURES, TEMPORAL_WINDOW, hidden_size_rec
        self.latent_length, self.n_layers, self.dropout = ZDIMS, 1, dropout_rec
        self.rnn_rec = nn.GRU(ZDIMS, hidden_size=hidden_size_rec, num_layers=1, bias=True, batch_first=True, dropout=dropout_rec,
                              bidirectional=False)
        self.hidden_to_output = nn.Linear(hidden_size_rec, NUM_FEATURES)

    def forward(self, inputs):
        """inputs = z tiled over time (B, T, Z) (rnn_model.py:311-312); the kernels read z = inputs[:, 0] once."""
        return self._run(inputs, inputs[:, 0, :], "dec")


class Decoder_Future_LEGACY(_DecoderBase):
    def __init__(self, TEMPORAL_WINDOW, ZDIMS, NUM_FEATURES, FUTURE_STEPS, hidden_size_pred, dropout_pred):
        super().__init__()
        self.num_features, self.future_steps, self.sequence_length = NUM_FEATURES, FUTURE_STEPS, TEMPORAL_WINDOW
        self.hidden_size, self.latent_length, self.n_layers, self.dropout = hidden_size_pred, ZDIMS, 1, dropout_pred
        self.rnn_pred = nn.GRU(ZDIMS, hidden_size=hidden_size_pred, num_layers=1, bias=True, batch_first=True, dropout=dropout_pred,
                               bidirectional=True)
        self.hidden_to_output = nn.Linear(hidden_size_pred * 2, NUM_FEATURES)

    def forward(self, inputs):
        return self._run(inputs, inputs[:, 0, :], "fut")


class RNN_VAE_LEGACY(RNN_VAE):
    _LEGACY = True

    def _build_modules(self, ZDIMS, NUM_FEATURES, FUTURE_DECODER, FUTURE_STEPS, h1, h2, h_rec, h_pred, d_enc, d_rec, d_pred, softplus):
        self.encoder = Encoder_LEGACY(NUM_FEATURES, h1, h2, d_enc)
        self.lmbda = Lambda_LEGACY(ZDIMS, h1, h2)
        self.decoder = Decoder_LEGACY(self.seq_len, ZDIMS, NUM_FEATURES, h_rec, d_rec)
        if FUTURE_DECODER:
            self.decoder_future = Decoder_Future_LEGACY(self.seq_len, ZDIMS, NUM_FEATURES, FUTURE_STEPS, h_pred, d_pred)


def _clone_state_dict(module, state_dict, prefix, local_metadata):
    # parameters are views into one flat bucket; hand out independent tensors like the reference's state_dict
    for k in list(state_dict.keys()):
        state_dict[k] = state_dict[k].clone()
    return state_dict
