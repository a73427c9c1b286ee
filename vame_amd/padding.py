"""Hidden sizes that are not a multiple of 32.

torch.nn.GRU takes any `hidden_size` (reference rnn_model.py:34,91,125: `hidden_size_layer_1`, `hidden_size_rec`,
`hidden_size_pred` come straight from config.yaml), while the gfx950 GRU kernels tile hidden units in blocks of 32.  For such a
model the parameters keep the reference's shapes (state_dict, checkpoints, optimizer state, the all-reduced gradient bucket) and the
kernels run on a ZERO-PADDED IMAGE of them: every hidden size is rounded up to the next multiple of 32 (of 64 between 256 and 512: pad32) and each block of H rows /
columns that indexes hidden units is laid out as H real entries followed by zeros.  A padded unit has zero input weights, zero
recurrent weights and zero biases, so r = u = 1/2, n = tanh(0) = 0 and h' = u h = 0 for all steps (its initial state is 0 too: the
padded rows of latent_to_hidden are zero), nothing downstream reads it (zero columns), and the real units see exactly the
reference's arithmetic.  The `hidden.view(2, B, H)` reinterpretation of the decoders (rnn_model.py:104,137) keeps its form: flat
offset (d*B + b)*H in the (B, 2H) buffer is row (d*B+b)//2, half (d*B+b)%2, which in the padded (B, 2H') buffer is (d*B + b)*H'.

`PadMap` holds the two index lists; `vame_index_copy_f32` moves parameters in (before a step) and gradients out (after backward).
"""
import torch

from .engine import ParamTable, Spec


def pad32(h):
    """The hidden size the kernels run a true size h on: the next multiple of 32 -- and, between 256 and 512, of 64: the persistent kernels of
    that range give a wave two 32-column blocks (gru_wide.hip: 320 / 384 / 448 / 512), and the per-step GEMM path that an odd multiple of 32
    would otherwise take is half as fast (288: 48 TF against 85 at 320 on one MI355X, profiles/r05_shape_table.txt) -- the 11-23 % of padded
    flops cost far less."""
    hp = (int(h) + 31) // 32 * 32
    if 256 < hp <= 512 and hp % 64:
        hp += 32
    return hp


def needs_padding(spec: Spec):
    return any(pad32(h) != h for h in (spec.H, spec.Hd, spec.Hf))


def _blocks(name, shape, spec: Spec):
    """(row block, col block) of hidden-unit indices for parameter `name`: the true block size h (padded to pad32(h)) or 0 when the
    dimension does not index hidden units."""
    if name.startswith("decoder_future."):
        h = spec.Hf
    elif name.startswith("decoder."):
        h = spec.Hd
    else:
        h = spec.H
    leaf = name.rsplit(".", 1)[-1]
    if leaf.startswith("weight_ih"):
        second_layer = "_l1" in leaf or ".rnn_2." in name                 # layer 1 of the encoder reads [fwd | bwd] of layer 0
        return h, (spec.H if second_layer else 0)
    if leaf.startswith("weight_hh"):
        return h, h
    if leaf.startswith("bias_ih") or leaf.startswith("bias_hh"):
        return h, 0
    if ".hidden_to_mean." in name or ".hidden_to_logvar." in name:
        return (0, h) if leaf == "weight" else (0, 0)
    if ".hidden_to_linear." in name:                                       # legacy Lambda: (4H, 4H), never used by forward
        return (h, h) if leaf == "weight" else (h, 0)
    if ".latent_to_hidden." in name:
        return h, 0
    if ".hidden_to_output." in name:
        return (0, h) if leaf == "weight" else (0, 0)
    raise KeyError(f"padding rule missing for parameter {name} {tuple(shape)}")


def _pad_index(n, h):
    """Positions of n consecutive true indices inside the padded dimension (blocks of h -> pad32(h))."""
    i = torch.arange(n, dtype=torch.int64)
    if not h or pad32(h) == h:
        return i, n
    assert n % h == 0, (n, h)
    hp = pad32(h)
    return (i // h) * hp + (i % h), n // h * hp


class PadMap:
    def __init__(self, spec: Spec, true_table: ParamTable, named_shapes, dev):
        self.true_spec = spec
        Hp, Hdp, Hfp = pad32(spec.H), pad32(spec.Hd), pad32(spec.Hf)
        self.spec = Spec(T=spec.T, F=spec.F, Z=spec.Z, H=Hp, FS=spec.FS, future=spec.future, softplus=spec.softplus, legacy=spec.legacy,
                         H_rec=0 if Hdp == Hp else Hdp, H_pred=0 if (not spec.future or Hfp == Hp) else Hfp, dropout=spec.dropout)
        padded_shapes, per = [], []
        for name, shape in named_shapes:
            rb, cb = _blocks(name, shape, spec)
            rows, R = _pad_index(shape[0], rb)
            if len(shape) == 2:
                cols, C = _pad_index(shape[1], cb)
                padded_shapes.append((name, (R, C)))
                per.append((name, rows, cols, C))
            else:
                padded_shapes.append((name, (R,)))
                per.append((name, rows, None, 1))
        self.table = ParamTable(padded_shapes)
        src, dst = [], []
        for (name, shape), (_, rows, cols, C) in zip(named_shapes, per):
            n = 1
            for v in shape:
                n *= v
            src.append(true_table.off(name) + torch.arange(n, dtype=torch.int64))
            if cols is None:
                dst.append(self.table.off(name) + rows)
            else:
                dst.append(self.table.off(name) + (rows[:, None] * C + cols[None, :]).reshape(-1))
        self.true_idx = torch.cat(src).to(dev)
        self.pad_idx = torch.cat(dst).to(dev)
        self.p = torch.zeros(self.table.numel, device=dev)
        self.g = torch.zeros(self.table.numel, device=dev)

    def push_params(self, flat_p):
        from . import ops
        ops.index_copy(self.p, self.pad_idx, flat_p, self.true_idx)

    def pull_grads(self, flat_g):
        from . import ops
        ops.index_copy(flat_g, self.true_idx, self.g, self.pad_idx)

    # activations that cross the module boundary with a hidden-unit axis: (.., k*H) <-> (.., k*H')
    def unpad_cols(self, t, k, h):
        hp = pad32(h)
        if hp == h:
            return t
        return t.reshape(*t.shape[:-1], k, hp)[..., :h].reshape(*t.shape[:-1], k * h).contiguous()

    def pad_cols(self, t, k, h, fill=0.0):
        hp = pad32(h)
        if hp == h:
            return t
        out = torch.full((*t.shape[:-1], k, hp), fill, device=t.device, dtype=t.dtype)
        out[..., :h] = t.reshape(*t.shape[:-1], k, h)
        return out.reshape(*t.shape[:-1], k * hp)
