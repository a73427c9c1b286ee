"""Host orchestration of the RNN-VAE forward / backward on MI355X.

The reference delegates this to torch autograd over nn.GRU / nn.Linear
(vame/model/rnn_model.py:162-179 forward, vame/model/rnn_vae.py:124-143 loss + backward).  Here
it is an explicit schedule of the C-ABI kernels (include/vame_hip.h) on the current HIP stream:

  forward   input-projection GEMMs -> GRU sequence kernels (2 or 4 (layer,dir) streams per
            launch) -> Lambda GEMMs + reparameterisation -> decoder GEMMs/GRUs -> output GEMMs
  loss      MSE(sum) x2 + KL(mean) + nuclear norm from the (Z,Z) Gram
  backward  the same graph reversed; weight gradients are split-K GEMMs over (batch x time)
            written straight into the flat gradient bucket that RCCL all-reduces.

All buffers are fp32 and live in a per-batch-size workspace (no allocation inside a step).
Sequences are stored as (B, T+2, 2H): slot 0 / T+1 hold the initial state of the forward /
reverse direction so h_{t-1} is a plain strided view for the BPTT kernels and the dW_hh GEMMs.
"""
from dataclasses import dataclass
from types import SimpleNamespace

import torch

from . import ops
from .ops import GB, GF, Operand

LOSS_REC, LOSS_FUT, LOSS_KLSUM, LOSS_KMEANS = 0, 1, 2, 3

# Side streams are per PROCESS and device, not per engine: a HIP stream is mapped to one of a few hardware queues when it is created, and
# which queue a step's side streams share with the caller's stream decides how its cross-stream waits resolve -- engines built one after the
# other in one process (bench legs, a training run followed by evaluation) measured 2.69 -> 2.75 -> 2.79 -> 2.85 ms per batch-256 step as
# each created its own five streams.  One set, created once, gives every engine the fresh-process placement.
_SIDE_STREAMS = {}


def _side_stream(dev, key):
    k = (dev.index, key)
    st = _SIDE_STREAMS.get(k)
    if st is None:
        st = _SIDE_STREAMS[k] = torch.cuda.Stream(device=dev)
    return st


@dataclass(frozen=True)
class Spec:
    T: int
    F: int
    Z: int
    H: int
    FS: int
    future: bool
    softplus: bool
    legacy: bool = False      # RNN_VAE_LEGACY topology (rnn_model.py:186-324): encoder = two stacked 1-layer bi-GRUs, softplus
                              # always, uni-directional reconstruction decoder, no latent->hidden initial states
    H_rec: int = 0            # hidden_size_rec / hidden_size_pred (rnn_model.py:148-160) when they differ from the encoder's
    H_pred: int = 0           # hidden size H (0 = same as H)
    dropout: float = 0.0      # dropout_encoder: inter-layer dropout of the 2-layer encoder GRU in training (rnn_model.py:34-35)

    @property
    def Hd(self):
        return self.H_rec or self.H

    @property
    def Hf(self):
        return self.H_pred or self.H


class ParamTable:
    """Name -> (offset, shape) layout of the flat fp32 parameter / gradient buffers."""

    def __init__(self, named_shapes):
        self.offsets, self.shapes, n = {}, {}, 0
        for name, shape in named_shapes:
            numel = 1
            for s in shape:
                numel *= s
            self.offsets[name], self.shapes[name] = n, tuple(shape)
            n += (numel + 3) // 4 * 4            # keep every tensor 16-byte aligned inside the bucket
        self.numel = n

    def off(self, name):
        return self.offsets[name]


class GruDir:
    """Packed per-(layer,direction) weights refreshed after every optimizer step."""

    def __init__(self, prefix, sfx, H, I, dev):
        self.w_ih, self.w_hh = f"{prefix}.weight_ih{sfx}", f"{prefix}.weight_hh{sfx}"
        self.b_ih, self.b_hh = f"{prefix}.bias_ih{sfx}", f"{prefix}.bias_hh{sfx}"
        self.H, self.I = H, I
        self.wp_x = torch.empty(3 * H * 32, device=dev) if (I <= 32 and I % 4 == 0) else None    # fused input projection
        self.wp_fwd = torch.empty(3 * H * H, device=dev)
        self.wp_bwd = torch.empty(3 * H * H, device=dev)
        self.bias_gi = torch.empty(3 * H, device=dev)
        self.b_hn = torch.empty(H, device=dev)


class Workspace:
    def __init__(self):
        self.t = {}

    def get(self, name, numel, dev, dtype=torch.float32, zero=False):
        t = self.t.get(name)
        if t is None or t.numel() < numel or t.device != dev:
            t = (torch.zeros if zero else torch.empty)(int(numel), device=dev, dtype=dtype)
            self.t[name] = t
            ops.ALLOC_GEN[0] += 1             # (a captured step graph holds the old address: see ops.ALLOC_GEN)
        return t


# Scheduling / kernel-choice options of an engine: CONSTRUCTOR ARGUMENTS (VAEEngine(..., options={...}); RNN_VAE(...).engine_options;
# the optional `vame_amd_engine:` mapping of config.yaml, vame_amd/model/rnn_vae.py) -- never the process environment.  The defaults are
# the measured best; the A/B tools set them by name (tools/step_ab.py B name=option:value,...; tools/shape_table.py SHAPE_ENGINE=option:value).
ENGINE_DEFAULTS = dict(
    group_wgrads=True,     # same-shape weight-gradient contractions leave as grouped launches (_group_wgrads)
    group_zproj=True,      # zdims <= 32: the decoders' <= 6 Linear layers of z (latent_to_hidden, W_ih on the time-constant input) as one launch
    fuse_heads=True,       # output Linear + MSE + dY + the Linear's weight gradient per decoder in ONE pass over the decoder states
                           # (vame_head_stream_f32, round 6) instead of four contractions + the MSE kernel; shapes it does not cover keep those
    small_streams=3,       # HIP streams for the independent small GEMMs before the decoders' launch
    wgrad_streams=0,       # 0 = auto: 4 streams up to batch 1024, ONE above (two give +1.5 % at batch 4096, but two large GEMMs sharing the
                           # chip each take twice as long, which makes per-kernel durations unreadable); 1 = caller's stream only
    coop=True,             # column-split GRU kernels for batches that leave most CUs idle (_coop_parts); False keeps the persistent ones
    coop_rounds=3,         # at most this many cooperative launches (row ranges) per stream set before the batch-tile-persistent kernels take over
    coop_cover=0,          # decoder + future decoder (four streams) as cooperative launches: 0 = whichever cover has fewer (launch x step) slots,
                           # 1 = all four streams per launch, 2 = one pair of directions per launch
    wide=True,             # 256 < H <= 512: persistent two-blocks-per-wave kernels (gru_wide.hip) instead of the per-step GEMM path
    wide_bwd=True,         # ... for BPTT too (False: step by step: per-step GEMM + gate kernel)
    nuc_side=True,         # nuclear-norm solve on a side stream next to the output heads
    bwd_overlap=True,      # the future decoder's two dW_hh contractions beside the small-kernel chain behind the decoders' BPTT launch
    skinny_side=False,     # narrow weight gradients (an output dimension <= 32) on a side stream beside the wide ones: measured +-0 at batch 256 /
                           # 1024 / 4096 in round 5 (tools/step_ab.py: 14.387 vs 14.392 ms) while it doubles both kernels' durations in every trace -> off
    split_wgrad=None,      # None = the f32-input matrix cores (default).  An int = the `opt` word of vame_gemm_group_bf16x6_f32 (0 = its
                           # defaults): the large grouped weight gradients (two k-major operands, N > 64, K >= 4096) run as the
                           # error-compensated split-bf16 contraction (bf16x6 planes, fp32 accumulate).  OPT-IN.
    split_proj=None,       # the same for the two large contractions with a row-major activation operand and K = a layer width: the second
                           # encoder layer's input projection and its data gradient (vame_gemm_bf16x6_f32; M >= 1024, N, K >= 128).  OPT-IN.
)


class VAEEngine:
    def __init__(self, spec: Spec, table: ParamTable, flat_p: torch.Tensor, flat_g: torch.Tensor, options=None):
        self.spec, self.table, self.p, self.g = spec, table, flat_p, flat_g
        opt = dict(ENGINE_DEFAULTS)
        unknown = set(options or ()) - set(opt)
        if unknown:
            raise ValueError(f"unknown engine option(s) {sorted(unknown)}; known: {sorted(opt)}")
        opt.update(options or {})
        self.dev = flat_p.device
        H, F, Z, Hd, Hf = spec.H, spec.F, spec.Z, spec.Hd, spec.Hf
        for hh in (H, Hd, Hf):
            if hh % 32 or hh > 4096:
                raise ValueError(f"hidden size {hh}: the gfx950 GRU kernels support multiples of 32 up to 4096 (other sizes: vame_amd.padding)")
        if spec.legacy and not (H == Hd == Hf):
            raise ValueError("RNN_VAE_LEGACY on the gfx950 kernels needs one hidden size for all GRUs")
        # H <= 256: persistent sequence kernels (h and the gate tiles stay on chip for all T steps).  Larger H: the gate
        # GEMM per step is a real dense contraction (M = batch) -> per-step vame_gemm_f32 + gate-math kernels
        self.force_stepwise = False
        if spec.legacy:       # same arithmetic as the 2-layer encoder, parameters live in two 1-layer modules
            e0, e1, s1 = "encoder.rnn_1", "encoder.rnn_2", "_l0"
        else:
            e0 = e1 = "encoder.encoder_rnn"
            s1 = "_l1"
        self.enc = [[GruDir(e0, "_l0", H, F, self.dev), GruDir(e0, "_l0_reverse", H, F, self.dev)],
                    [GruDir(e1, s1, H, 2 * H, self.dev), GruDir(e1, s1 + "_reverse", H, 2 * H, self.dev)]]
        self.dec = [GruDir("decoder.rnn_rec", "_l0", Hd, Z, self.dev)]
        if not spec.legacy:
            self.dec.append(GruDir("decoder.rnn_rec", "_l0_reverse", Hd, Z, self.dev))
        self.h0_from_z = not spec.legacy      # Linear(z).view(2,B,H) initial states (rnn_model.py:103-104,136-137); legacy: zeros
        self.fut = ([GruDir("decoder_future.rnn_pred", "_l0", Hf, Z, self.dev),
                     GruDir("decoder_future.rnn_pred", "_l0_reverse", Hf, Z, self.dev)] if spec.future else [])
        self.ws = Workspace()
        self._wgrad_queue = None
        self.group_wgrads = bool(opt["group_wgrads"])
        self.group_zproj = bool(opt["group_zproj"])
        self.fuse_heads = bool(opt["fuse_heads"])          # (the parity tests run the step both ways)
        self._heads_deferred = False
        self._B_bwd = None
        self._side_streams = []
        self.small_streams = int(opt["small_streams"])
        self.wgrad_streams = int(opt["wgrad_streams"])
        self.coop = bool(opt["coop"])
        self.coop_rounds = int(opt["coop_rounds"])
        self.coop_cover = int(opt["coop_cover"])
        # (these arrive from the user's config.yaml, `vame_amd_engine:` -- refuse values that would only fail deep inside a launch plan)
        for name, ok, want in (("coop_cover", self.coop_cover in (0, 1, 2), "0 (by cost), 1 (all streams per launch) or 2 (pairs of directions)"),
                               ("coop_rounds", self.coop_rounds >= 1, ">= 1"), ("small_streams", self.small_streams >= 0, ">= 0"),
                               ("wgrad_streams", self.wgrad_streams >= 0, ">= 0")):
            if not ok:
                raise ValueError(f"engine option {name} = {opt[name]!r}: expected {want}")
        self.wide = bool(opt["wide"])
        self.wide_bwd = bool(opt["wide_bwd"])
        self.split_wgrad = None if opt["split_wgrad"] is None else int(opt["split_wgrad"])
        self.split_proj = None if opt["split_proj"] is None else int(opt["split_proj"])
        # BPTT / forward kernel choice of the persistent H <= 256 launches (ops.KERNEL_*): an argument of every launch (descriptor field),
        # AUTO = the library's measured default per hidden size; the A/B tests set these attributes
        self.gru_bwd_kernel = ops.KERNEL_AUTO
        self.gru_fwd_kernel = ops.KERNEL_AUTO
        self._coop_state = None
        self._nuc_state = None
        # nuclear-norm solve (one workgroup, ~0.2 ms, nothing else on the chip) on a side stream NEXT TO THE OUTPUT HEADS: it starts
        # when the decoders' GRU launch has finished and runs beside the HBM-bound head / MSE / dY kernels, which leave most of a
        # CU's registers and LDS free -- unlike the GRU launches, which need whole CUs (a solve beside them was measured to cost
        # as much as it saves, profiles/NOTES.md section 8).  Joined before dz reads Minv and before the loss terms are handed out.
        # (wide shapes, H > 256: the side streams were measured to COST time -- configs[3], batch 8192: 202.6 ms per step with them, 201.5 without,
        # profiles/r06_cfg3_overlap_ab.txt -- their partners there are the LDS-filling wide GRU launches and contractions that already fill the chip)
        wide_shape = max(H, Hd, Hf) > 256
        self.nuc_side = bool(opt["nuc_side"]) and not wide_shape
        # Small batches: the decoders' BPTT launch is the cooperative column-split kernel, one workgroup on EVERY CU, and the solve's single
        # workgroup still holds a CU when it starts (their LDS footprints exclude each other: 153 KB + 98 KB), so one 8-member group of the
        # launch starts late.  Measured at batch 256 (tools/step_ab.py, profiles/r04_b256_overlap.txt): joining the solve in front of that
        # launch 2.873 ms/step, leaving them side by side 2.790, no side stream at all 2.961 -- the late group costs less than the
        # serialisation, and its members' waits are bounded polls of 0.3 s against a 0.2 ms solve.  True = join (diagnostics).
        self.nuc_join_before_coop = False
        # the future decoder's two dW_hh contractions beside the small-kernel chain that follows the decoders' BPTT launch (_early_wgrads)
        self.bwd_overlap = bool(opt["bwd_overlap"]) and not wide_shape
        self.skinny_side = bool(opt["skinny_side"])
        self._wgrad_plans = {}
        self._coop_covers = {}
        self.wgrad_min_rounds = 0          # grouped weight gradients: whole rounds of 3 workgroups per CU, at least this many (0 = by K, see _group_wgrads)
        self._early_stream, self._early_pending = None, False
        self._nuc_stream = None
        self._nuc_pending = None          # arguments of a deferred cluster_terms()
        self._nuc_event = None            # recorded on the side stream after the solve
        self._drop_mask = None
        self._rng = None                  # device state of the reparameterisation's N(0,1) stream (seed_rng)
        self.capturing = False            # inside a hipGraph stream capture (rnn_vae.GraphedTrainStep): no event queries, no host copies
        self._eps_used = None
        self.packed_version = -1
        self.version = 0          # bumped by the owner whenever flat_p changes
        self.serial = 0           # bumped by every call that overwrites the forward workspace (stale-backward guard)
        self._B = None

    # ------------------------------------------------------------------ helpers
    @property
    def stepwise(self):
        """True when the ENCODER runs the per-step GEMM path (tests force it on small models by assignment)."""
        return self._stepwise(self.spec.H)

    @stepwise.setter
    def stepwise(self, v):
        self.force_stepwise = bool(v)

    def _stepwise(self, H):
        return self.force_stepwise or H > 256

    def _wide(self, H):
        """256 < H <= 512: forward recurrence and (unless wide_bwd is off) BPTT run in the two-blocks-per-wave persistent kernels of
        gru_wide.hip (fragment-order stash); wide_bwd = False: BPTT step by step (per-step GEMM + gate kernel reading that stash)."""
        return self.wide and not self.force_stepwise and H > 256 and ops.gru_wide_supported(H)

    def _all_dirs(self):
        return self.enc[0] + self.enc[1] + self.dec + self.fut

    def repack(self):
        if self.packed_version == self.version:
            return
        ops.gru_pack_batch([(self._pv(d.w_hh), self._pv(d.b_ih), self._pv(d.b_hh), d.H, d.wp_fwd, d.wp_bwd, d.bias_gi, d.b_hn,
                             self._pv(d.w_ih) if d.wp_x is not None else None, d.I, d.wp_x) for d in self._all_dirs()])
        self.packed_version = self.version

    def _pv(self, name):
        o = self.table.off(name)
        n = 1
        for s in self.table.shapes[name]:
            n *= s
        return self.p[o:o + n]

    def P(self, name, ld):
        return Operand(self.p, ld, off=self.table.off(name))

    def buf(self, name, *shape, zero=False):
        n = 1
        for s in shape:
            n *= s
        return self.ws.get(name, n, self.dev, zero=zero)

    def _early_wgrads(self, after_event, pick):
        """Issue the queued weight-gradient GEMMs selected by `pick(job)` NOW, on a side stream that starts at `after_event`
        (recorded behind the BPTT launch that produced their operands).  Used for ONE MFMA-bound contraction next to the chain of
        small / HBM-bound kernels between the decoders' BPTT launch and the encoder's (time sums, dz, Lambda backward: ~0.4 ms in
        which most CUs idle); the caller joins the stream before the next GRU launch, which wants every CU for itself."""
        mine = [j for j in self._wgrad_queue if pick(j)]
        if len(mine) < 1:
            return
        self._wgrad_queue = [j for j in self._wgrad_queue if not pick(j)]
        if self._early_stream is None:
            self._early_stream = _side_stream(self.dev, "early")
        self._early_stream.wait_event(after_event)
        with torch.cuda.stream(self._early_stream):
            for j in self._group_wgrads(mine, ws_name="splitk_early"):
                self._gemm_wgrad_now(*j, lane="_early")
        self._early_pending = True

    def _join_early(self):
        if self._early_pending:
            torch.cuda.current_stream(self.dev).wait_stream(self._early_stream)
            self._early_pending = False

    def _group_wgrads(self, jobs, ws_name="splitk_group", after_first=None, deferred=None):
        """Weight-gradient GEMMs of one shape and operand layout -- the dW_hh of the six
        T-step (layer, direction) streams, the two layer-1 dW_ih, the future decoder's two dW_hh -- go out as ONE grouped launch
        each (vame_gemm_group_f32): all their k-slabs are dealt to the XCDs together, so nothing idles at the boundary between
        them and each problem needs fewer partial sums for the same occupancy.  Returns the jobs left for single launches."""
        if not self.group_wgrads:
            return jobs
        # the grouping depends on shapes and layouts only: planned once per job signature (a quarter of backward()'s host time otherwise)
        sig = (ws_name, self.wgrad_min_rounds, tuple((j[0], j[1], j[2], j[3].ld, j[3].seg, j[3].seg_stride, j[4].ld, j[4].seg, j[4].seg_stride,
                                                      j[5], j[6], j[7], j[8]) for j in jobs))
        plan = self._wgrad_plans.get(sig)
        if plan is None:
            plan = self._wgrad_plans[sig] = self._plan_wgrads(jobs)
        launches, rest_idx = plan

        def launcher(grp, idx, M, N, K, sk, gap_at, gap, c_offs):
            def run(ws_name_=ws_name):
                ws = self.ws.get(ws_name_, len(idx) * sk * M * N, self.dev)
                As, Bs = [jobs[i][3] for i in idx], [jobs[i][4] for i in idx]
                split = self.split_wgrad if (self.split_wgrad is not None and N > 64 and M >= 128 and K >= 4096
                                             and ops.gemm_split_ok(M, N, K, As, Bs, sk, gap_at, gap)) else None
                ops.gemm_group(M, N, K, As, 1, Bs, 1, self.g, c_offs, N, sk, ws, a_gap_at=gap_at, a_gap=gap, split=split)
            return (float(M) * N * K * len(idx), run)
        if deferred is not None:               # the caller spreads the grouped launches over its streams together with the single ones
            deferred += [launcher(*L) for L in launches]
            return [jobs[i] for i in rest_idx]
        for L in launches:
            if L[0] == 1 and after_first is not None:
                after_first()
                after_first = None
            launcher(*L)[1]()
        return [jobs[i] for i in rest_idx]

    def _plan_wgrads(self, jobs):
        """([(group ordinal, job indices, M, N, K, split-K, gap_at, gap, output offsets)] in launch order, indices of the jobs left over)."""
        groups, rest = {}, []
        for i, j in enumerate(jobs):
            M, N, K, A, Bop, gname, row_off, gap_at, gap = j
            # large K: (M <= 32: the 24/30-row heads keep their own skinny-M tile).  Small K (K = batch <= 2048: the contractions
            # over the batch alone -- decoders' dW_ih, latent_to_hidden, Lambda): every such launch is a chain of <= 8 dependent k-tiles
            # on a handful of workgroups, 12-17 us of latency each whatever its size, so same-shape ones leave as one grouped launch
            # of single-k-tile workgroups + one reduction, any M
            if row_off == 0 and ((M > 32 and K >= 8 * 256) or (256 <= K < 8 * 256 and _slabs(K, 8) >= 8)):
                key = (M, N, K, A.ld, A.seg, A.seg_stride, Bop.ld, Bop.seg, Bop.seg_stride, gap_at, gap)
                groups.setdefault(key, []).append(i)
            else:
                rest.append(i)
        launches, launched = [], 0
        # largest group first (the six T-step dW_hh contractions: the step's dominant launch); `after_first` runs behind it
        for key, members in sorted(groups.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[0][2] * len(kv[1])):
            if len(members) < 2:
                rest += members
                continue
            grp = launched
            launched += 1
            M, N, K = key[:3]
            for i0 in range(0, len(members), 8):
                part = members[i0:i0 + 8]
                if len(part) < 2:
                    rest += part
                    continue
                tiles = ((M + 127) // 128) * ((N + 127) // 128 if N > 64 else 1) * len(part)
                # whole rounds of 3 workgroups per CU: the smallest multiple of 8 with enough rounds, capped so that a workgroup keeps >= 8 k-tiles
                cands = [k for k in range(8, 129, 8) if k * 8 * 32 <= K]
                # one whole round where the k-slabs stay short (K = batch x time < 2^18: +0.5 % at batch 4096, +1.7 % at 256 against two rounds --
                # half the partial sums), two where a slab is long enough for the workgroups to drift apart (K = 491,520: one round -0.4 %);
                # profiles/r04_wgrad_rounds.txt
                rounds = self.wgrad_min_rounds or (1 if K < (1 << 18) else 2)
                full = [k for k in cands if tiles * k >= rounds * 768 and (tiles * k) % 768 == 0]
                sk = full[0] if full else next((k for k in cands if tiles * k >= rounds * 768), cands[-1] if cands else 8)
                if K >= (1 << 18) and full:
                    # long slabs (configs[3]: K = 491,520, 288 tiles per slab): the workgroups that share an operand panel drift apart and re-fetch it -- 56.9 GB per
                    # launch against 24.2 GB of operands at split-K 8, 47.7 GB at 96 for the same time within 1 % (profiles/r05_cfg3_gemm_splitk_traffic.txt):
                    # the shortest slabs that still fill whole rounds, up to 96
                    sk = max(k for k in full if k <= 96)
                launches.append((grp, part, M, N, K, sk, key[9], key[10], [self.table.off(jobs[i][5]) for i in part]))
        return launches, rest

    def _sum_into(self, C, M, N, jobs):
        """C (M,N) = sum of A_j (M,K_j) @ B_j (K_j,N).  Products of one K go out as one grouped launch whose partial sums are reduced
        straight into C (vame_gemm_group_f32 with a shared output): two launches instead of two per product."""
        first = True
        by_k = {}
        for K, A, Bop in jobs:
            by_k.setdefault((K, A.ld, Bop.ld), []).append((A, Bop))
        for (K, _, _), members in by_k.items():
            kper = -(-(-(-K // 8)) // 32) * 32
            if self.group_wgrads and 2 <= len(members) <= 8 and -(-K // kper) >= 8:
                ws = self.ws.get("splitk_sum", len(members) * 8 * M * N, self.dev)
                ops.gemm_group(M, N, K, [m[0] for m in members], 0, [m[1] for m in members], 1, C, [0] * len(members), N, 8, ws,
                               accumulate=not first)
                first = False
            else:
                for A, Bop in members:
                    ops.gemm(M, N, K, A, 0, Bop, 1, C, N, accumulate=not first, splitk=0)
                    first = False

    def _splitk(self, M, N, K):
        # ~3 workgroups per CU on all 256 CUs; with split-K >= 8 a whole k-slab lives on one XCD (gemm.hip
        # map_tile), so keep split-K a multiple of 8 to load the 8 XCDs evenly
        tiles = ((M + 127) // 128) * ((N + 127) // 128 if N > 64 else 1)
        sk = max(1, min(768 // max(tiles, 1), K // 256))
        if sk >= 8:
            sk = sk // 8 * 8
        return sk

    def _gemm_wgrad(self, M, N, K, A, B, gname, row_off=0, gap_at=0, gap=0, lane=0):
        """flat_g[gname][row_off:row_off+M, :N] = A^T B with split-K over K = batch x time.  Inside backward() the call is
        queued (nothing reads flat_g before the optimizer) and issued by _flush_wgrads(); `lane` picks the split-K scratch."""
        if self._wgrad_queue is not None:
            self._wgrad_queue.append((M, N, K, A, B, gname, row_off, gap_at, gap))
            return
        self._gemm_wgrad_now(M, N, K, A, B, gname, row_off, gap_at, gap, lane)

    def _gemm_wgrad_now(self, M, N, K, A, B, gname, row_off=0, gap_at=0, gap=0, lane=0):
        sk = self._splitk(M, N, K)
        ws = self.ws.get(f"splitk{lane}", max(sk * M * N, 1), self.dev) if sk > 1 else None
        ops.gemm(M, N, K, A, 1, B, 1, self.g, N, c_off=self.table.off(gname) + row_off * N, splitk=sk, ws=ws,
                 a_gap_at=gap_at, a_gap=gap)

    def _parallel(self, jobs, n):
        """Run independent launch closures on up to n streams (round-robin), joined before returning.  Only for groups of
        kernels that are not next to a GRU launch (see _flush_wgrads) and share no scratch."""
        if self.dev.type != "cuda" or n < 2 or len(jobs) < 2:
            for j in jobs:
                j()
            return
        n = min(n, len(jobs))
        while len(self._side_streams) < n - 1:
            self._side_streams.append(_side_stream(self.dev, len(self._side_streams)))
        main = torch.cuda.current_stream(self.dev)
        sides = self._side_streams[:n - 1]
        for sd in sides:
            sd.wait_stream(main)
        for i, j in enumerate(jobs):
            k = i % n
            if k == 0:
                j()
            else:
                with torch.cuda.stream(sides[k - 1]):
                    j()
        for sd in sides:
            main.wait_stream(sd)

    def _flush_wgrads(self):
        """Issue the queued weight-gradient GEMMs.  They are independent of each other, so on the GPU they go out on
        several streams (see `wgrad_streams`), largest first: the tail of one (workgroups finish up to 5 % apart) and its split-K reduction overlap the next
        one's body instead of idling the chip at ~20 kernel boundaries.  They run after the last GRU launch of the backward
        pass on purpose -- next to a GRU launch they would take LDS from workgroups that need a whole CU."""
        jobs, self._wgrad_queue = self._wgrad_queue, None
        if not jobs:
            return
        n = (self.wgrad_streams or (4 if (self._B_bwd or 0) <= 1024 else 1)) if self.dev.type == "cuda" else 1
        skinny = []
        if self.skinny_side and n == 1 and self.dev.type == "cuda":
            # large batches: the narrow contractions (an output dimension <= 32: the layer-0 / decoder dW_ih, the output heads', Lambda's)
            # stream their K = batch x time operand at HBM speed with the matrix pipes nearly idle, the wide ones are MFMA-bound with
            # HBM nearly idle -> the narrow ones run on a side stream beside the wide ones, starting BEHIND the dominant grouped launch
            # (which keeps the chip to itself, so its duration stays a statement about that kernel)
            skinny = [j for j in jobs if min(j[0], j[1]) <= 32]
            jobs = [j for j in jobs if min(j[0], j[1]) > 32]

        def start_skinny():
            if not skinny:
                return
            if self._early_stream is None:
                self._early_stream = _side_stream(self.dev, "early")
            self._early_stream.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(self._early_stream):
                for j in self._group_wgrads(skinny, ws_name="splitk_skinny"):
                    self._gemm_wgrad_now(*j, lane="_sk")
            self._early_pending = True
        # several streams and a batch up to 512: the grouped launches are spread over the streams as well -- at batch 256 each is 0.75-1.5 rounds
        # of workgroups with a tail (180 + 106 us back to back), side by side they fill each other's (batch 128 / 256 / 384: +5 / +3 / +2 % of the
        # step with the bias sums' move; at 768 / 1024 two such launches sharing the chip cost 1.5-2 %, so there they stay in sequence)
        grouped = [] if (n >= 2 and (self._B_bwd or 0) <= 512) else None
        jobs = self._group_wgrads(jobs, after_first=start_skinny, deferred=grouped)   # same-shape contractions leave as grouped launches first
        if skinny and not self._early_pending:
            start_skinny()
        if skinny:
            for j in jobs:
                self._gemm_wgrad(*j)
            self._join_early()
            return
        if n < 2 or len(jobs) + len(grouped or ()) < 2:
            for _, run in (grouped or ()):
                run()
            for j in jobs:
                self._gemm_wgrad(*j)
            return
        while len(self._side_streams) < n - 1:
            self._side_streams.append(_side_stream(self.dev, len(self._side_streams)))
        main = torch.cuda.current_stream(self.dev)
        items = list(grouped or ()) + [(float(j[0]) * j[1] * j[2], j) for j in jobs]
        items.sort(key=lambda it: -it[0])
        lanes, load = [[] for _ in range(n)], [0.0] * n
        for fl, it in items:                             # greedy balance by flops
            k = load.index(min(load))
            lanes[k].append(it)
            load[k] += fl

        def run_lane(k):
            for it in lanes[k]:
                if callable(it):
                    it(f"splitk_group_l{k}")             # (a lane's launches are ordered: they share the lane's split-K scratch)
                else:
                    self._gemm_wgrad(*it, lane=k)
        for side in self._side_streams[:n - 1]:
            side.wait_stream(main)
        run_lane(0)
        for k, side in enumerate(self._side_streams[:n - 1], start=1):
            with torch.cuda.stream(side):
                run_lane(k)
        for side in self._side_streams[:n - 1]:
            main.wait_stream(side)

    # ------------------------------------------------------------------ GRU sequence dispatch
    def _multi_rank(self):
        return torch.distributed.is_available() and torch.distributed.is_initialized()

    def check_async_errors(self, all_ranks=False):
        """Called where the host synchronises anyway (end of an epoch, before a checkpoint, after an inference call whose result
        goes to the host): surfaces device-side failures that cannot raise.  Free when no cooperative launch happened since.
        all_ranks (the epoch-end checks of train() / test(): every rank is at this program point): the status words are MAX-reduced
        over the ranks first, so either every rank raises or none does -- a rank raising on its own would leave the others in
        their next collective."""
        reduce = None
        if all_ranks and self._multi_rank():
            def reduce(n):
                t = torch.tensor([int(n)], dtype=torch.int32, device=self.dev)
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
                return int(t.item())
        if self._coop_state is None:
            # no cooperative launch on THIS rank so far (a smaller last batch, engine.coop off here): the collective still has to be entered,
            # contributing 0 -- a rank that skipped it would leave the others waiting at the end of the epoch
            if reduce is not None and reduce(0):
                raise ops._lib.VameHipError("cooperative GRU kernel: a hand-off wait timed out on another rank; the affected optimizer step was dropped on every rank")
            return
        self._coop_state.check(reduce=reduce)

    def poll_async_errors(self):
        """Start of a step: raise if an earlier step's status snapshot has arrived non-zero (never blocks)."""
        if self._coop_state is not None and not self.capturing:
            self._coop_state.poll()

    def snapshot_async_errors(self, reduced=False):
        """End of a step: enqueue the 4-byte status copy that poll_async_errors() of the next steps looks at.  With several ranks
        the word to look at is the all-reduced one (share_status), so the copy is enqueued by allreduce_gradients (reduced=True) and
        the per-rank call at the end of loss_step does nothing."""
        # (under a process group the rank-local call does nothing from the very first step on -- `shared` is only set by the first
        # all-reduce, and a rank-local word in the snapshot ring would let ONE rank raise a step later)
        if self.capturing:           # (a step being captured into a graph: the copy to the host and its event are the replayer's business)
            return
        if self._coop_state is not None and (reduced or (self._coop_state.shared is None and not self._multi_rank())):
            self._coop_state.snapshot()

    def abort_flag(self):
        """Device word the optimizer kernel tests before it touches the weights (None: no cooperative launch so far)."""
        if self._coop_state is None:
            return None
        return self._coop_state.status if self._coop_state.shared is None else self._coop_state.shared

    def unshare_status(self):
        if self._coop_state is not None and self._coop_state.shared is not None:
            self._coop_state.shared = None

    def share_status(self, word):
        """Several ranks: copy this rank's status into `word` (one float behind the gradient bucket) so that the gradient all-reduce
        carries it; from then on `word` is the abort flag and the snapshot source on every rank (ops.CoopState)."""
        if self._coop_state is not None:
            self._coop_state.shared = word
            word.copy_(self._coop_state.status)

    def _split_rows(self, M, N, K, A, Bop, b_kmajor):
        """The `opt` word of vame_gemm_bf16x6_f32 for this row-major-A contraction (option split_proj), or None = the f32-input kernel:
        not opted in, too small to pay, or a layout the split form does not take (ops.gemm_split_rows_ok)."""
        if self.split_proj is None or M < 1024 or N < 128 or K < 128 or not ops.gemm_split_rows_ok(M, N, K, A, 0, Bop, b_kmajor):
            return None
        return self.split_proj

    def _coop_parts(self, rows, B, tkey=None, H=None):
        """[(streams, (row0, nrows))] cooperative launches that cover this GRU launch, each fitting one workgroup per CU; at
        most two per stream set (beyond that the persistent kernels win).  [] = use the persistent kernels."""
        H = H or self.spec.H
        if self._stepwise(H) or not self.coop:
            return []
        steps = lambda r: int(r[tkey]) if (r and tkey is not None) else 1                    # noqa: E731
        # the cover depends on (stream count, their lengths, batch, hidden size) only: decided once per signature
        sig = (len(rows), tuple(steps(r) for r in rows), B, H, self.coop_rounds, self.coop_cover)
        cover = self._coop_covers.get(sig)
        if cover is None:
            idx = list(range(len(rows)))
            options = [([idx], ops.coop_row_chunks(len(rows), B, H, self.coop_rounds))]
            if len(rows) > 2 and len(rows) % 2 == 0:                   # decoder + future decoder: one pair of directions each
                options.append(([idx[i:i + 2] for i in range(0, len(rows), 2)], ops.coop_row_chunks(2, B, H, self.coop_rounds)))
            options = [(sets, ch) for sets, ch in options if ch]
            if self.coop_cover and len(options) > 1:
                options = [options[self.coop_cover - 1]]
            if options:
                # every launch lasts as long as its longest sequence: pick the cover with the fewest (launch x step) slots; on a tie the pairs
                # (their last row range is more often small enough for 16-row groups: batch 768 4.77 -> 4.65 ms per step)
                sets, chunks = min(reversed(options), key=lambda o: len(o[1]) * sum(max(steps(rows[i]) for i in st) for st in o[0]))
                cover = [(st, ch) for st in sets for ch in chunks]
            else:
                cover = []
            self._coop_covers[sig] = cover
        if not cover:
            return []
        if self._coop_state is None:
            self._coop_state = ops.CoopState(self.dev)
        return [([rows[i] for i in st], ch) for st, ch in cover]

    def _gru_fwd(self, rows, B):
        """One launch for streams of one hidden size; streams of different sizes (hidden_size_rec != hidden_size_pred) go out
        as one launch per size."""
        for H in sorted({r["_s"].d.H for r in rows}, reverse=True):
            part_rows = [r for r in rows if r["_s"].d.H == H]
            if self._wide(H):
                for i in range(0, len(part_rows), 2):                # two streams per launch: one XCD parity class each
                    ops.gru_wide_fwd(part_rows[i:i + 2], B, H, kernel=self.gru_fwd_kernel if (self.gru_fwd_kernel != ops.KERNEL_SKEWED or H % 128 == 0) else ops.KERNEL_AUTO)
                continue
            if self._stepwise(H):
                for r in part_rows:
                    self._stepwise_fwd(r["_s"], B)
                continue
            if all(r[GF["Y"]] and not r.get(GF["WPX"]) for r in part_rows):
                parts = self._coop_parts(part_rows, B, GF["T"], H)
                for part, chunk in parts:
                    ops.gru_coop_fwd(part, B, H, self._coop_state, rows=chunk)
                if parts:
                    continue
            k = self.gru_fwd_kernel
            ops.gru_seq_fwd(part_rows, B, H, kernel=k if ops.gru_seq_fwd_has_kernel(H, k) else ops.KERNEL_AUTO)

    def _gru_bwd(self, rows, B):
        for H in sorted({r["_s"].d.H for r in rows}, reverse=True):
            part_rows = [r for r in rows if r["_s"].d.H == H]
            if self._wide(H) and self.wide_bwd:
                for i in range(0, len(part_rows), 2):
                    ops.gru_wide_bwd(part_rows[i:i + 2], B, H)
                continue
            if self._stepwise(H):
                for r in part_rows:
                    self._stepwise_bwd(r["_s"], B)
                continue
            parts = self._coop_parts(part_rows, B, GB["T"], H)
            for part, chunk in parts:
                ops.gru_coop_bwd(part, B, H, self._coop_state, rows=chunk)
            if not parts:
                k = self.gru_bwd_kernel
                if k == ops.KERNEL_WS and not ops.gru_seq_bwd_has_kernel(H, k):
                    k = ops.KERNEL_AUTO
                ops.gru_seq_bwd(part_rows, B, H, kernel=k)

    def _stepwise_fwd(self, s, B):
        """One (layer,direction) stream step by step: gh = h_{t-1} W_hh^T (GEMM) then the gate kernel."""
        d, T = s.d, s.T
        H = d.H
        Yrow, col = (s.y_T + 2) * 2 * H, s.dirn * H
        Y = s.Y if (s.Y is not None and s.write_y) else self.buf("Y_step", B, s.y_T + 2, 2 * H)
        yv = Y[:B * Yrow].view(B, s.y_T + 2, 2 * H)
        slot0 = T + 1 if s.dirn else 0                            # padded slot holding the initial state
        if s.h0 is not None:
            yv[:, slot0, col:col + H].copy_(s.h0[s.h0_off:s.h0_off + B * H].view(B, H))
        else:
            yv[:, slot0, col:col + H].zero_()
        gh = self.buf("gh_step", B, 3 * H)
        for step in range(T):
            t = T - 1 - step if s.dirn else step
            hp_off = ((t + 2) if s.dirn else t) * 2 * H + col     # previous state: time t-1 (fwd) / t+1 (reverse)
            ops.gemm(B, 3 * H, H, Operand(Y, Yrow, off=hp_off), 0, self.P(d.w_hh, H), 0, gh, 3 * H)
            ops.gru_cell_fwd(s.gi, t * s.gi_t, s.gi_row, gh, d.b_hn, Y, hp_off, Yrow, Y, (t + 1) * 2 * H + col, Yrow,
                             s.stash, t * 5 * H if s.stash is not None else 0, T * 5 * H, B, H)
        if s.hn is not None:
            last = 1 if s.dirn else T
            rows = s.hn.numel() // s.hn_row
            s.hn[:rows * s.hn_row].view(rows, s.hn_row)[:B, s.hn_off:s.hn_off + H].copy_(yv[:, last, col:col + H])

    def _stepwise_bwd(self, s, B):
        d, T = s.d, s.T
        H = d.H
        dh, dgh = self.buf("dh_step", B, H), self.buf("dgh_step", B, 3 * H)
        dhv = dh[:B * H].view(B, H)
        if s.dhn is not None:
            rows = s.dhn.numel() // s.dhn_row
            dhv.copy_(s.dhn[:rows * s.dhn_row].view(rows, s.dhn_row)[:B, s.dhn_off:s.dhn_off + H])
        else:
            dhv.zero_()
        # (B x H) has few 128x128 tiles: split K = 3H three ways so the per-step GEMM fills the 256 CUs
        sk_b = 3 if ((B + 127) // 128) * ((H + 127) // 128) < 512 else 1
        ws_b = self.ws.get("splitk_step", sk_b * B * H, self.dev) if sk_b > 1 else None
        for step in range(T):
            fstep = T - 1 - step
            t = T - 1 - fstep if s.dirn else fstep
            if self._wide(H):
                ops.gru_cell_bwd_frag(s.stash, T, t, dh, s.dY, (t * 2 * H + s.dirn * H) if s.dY is not None else 0, s.dy_T * 2 * H, s.dG, t * 4 * H,
                                      T * 4 * H, dgh, B, H)
            else:
                ops.gru_cell_bwd(s.stash, t * 5 * H, T * 5 * H, dh, s.dY, (t * 2 * H + s.dirn * H) if s.dY is not None else 0,
                                 s.dy_T * 2 * H, s.dG, t * 4 * H, T * 4 * H, dgh, B, H)
            ops.gemm(B, H, 3 * H, Operand(dgh, 3 * H), 0, self.P(d.w_hh, H), 1, dh, H, accumulate=True, splitk=sk_b, ws=ws_b)
        if s.dh0 is not None:
            s.dh0[s.dh0_off:s.dh0_off + B * H].view(B, H).copy_(dhv)
        ntiles = (B + 31) // 32                                   # bias partials: everything in row 0 of the (ntiles,4H) buffer
        s.dbias[:ntiles * 4 * H].zero_()
        ops.colsum(s.dG, 0, B * T, 4 * H, 4 * H, s.dbias, 0)

    # ------------------------------------------------------------------ forward
    def _gru_fwd_stream(self, d: GruDir, gi, gi_row, gi_t, h0, h0_off, Y, y_cols, y_T, dirn, hn, hn_off, hn_row, stash, T,
                        write_y=True):
        H = d.H
        spec = SimpleNamespace(d=d, gi=gi, gi_row=gi_row, gi_t=gi_t, h0=h0, h0_off=h0_off, Y=Y, y_T=y_T, dirn=dirn, hn=hn,
                               hn_off=hn_off, hn_row=hn_row, stash=stash, T=T, write_y=write_y)
        return {"_s": spec,
                GF["GI"]: ops.addr(gi), GF["GI_ROW"]: gi_row, GF["GI_T"]: gi_t, GF["WP"]: ops.addr(d.wp_fwd),
                GF["BHN"]: ops.addr(d.b_hn), GF["H0"]: ops.addr(h0, h0_off) if h0 is not None else 0, GF["H0_ROW"]: H,
                GF["Y"]: ops.addr(Y, y_cols + dirn * H) if (write_y and Y is not None) else 0,
                GF["Y_ROW"]: (y_T + 2) * 2 * H, GF["Y_T"]: 2 * H,
                GF["HN"]: ops.addr(hn, hn_off) if hn is not None else 0, GF["HN_ROW"]: hn_row,
                GF["STASH"]: ops.addr(stash) if stash is not None else 0, GF["T"]: T, GF["REVERSE"]: dirn, GF["PAD"]: 1}

    def encode(self, win, win_row, B, training, drop_mask=None):
        """win: (B, >=T, F) windows with row stride win_row (elements).  Returns hn (B,4H).
        drop_mask (B,T,2H) of {0,1}: inter-layer dropout of the encoder (training, spec.dropout > 0): layer 1 reads
        Y0 * mask / (1 - p) (torch.nn.GRU semantics: the final states of layer 0 stay undropped)."""
        s, H, T, F = self.spec, self.spec.H, self.spec.T, self.spec.F
        self.repack()
        self.serial += 1
        x_op = Operand(win, F, seg=T, seg_stride=win_row)
        Y0 = self.buf("Y0", B, T + 2, 2 * H)
        hn = self.buf("hn", B, 4 * H)
        rows = []
        # layer 0: F <= 32 features -> the input projection runs inside the sequence kernel on the window tile itself
        coop = bool(self._coop_parts([None, None], B)) if not self.stepwise else False
        fused = (not self.stepwise) and (not coop) and self.enc[0][0].wp_x is not None and win_row % 4 == 0 and win.data_ptr() % 16 == 0
        for dirn, d in enumerate(self.enc[0]):
            st = self.buf(f"st_e0_{dirn}", ops.gru_stash_floats(B, T, H)) if training else None
            if fused:
                row = self._gru_fwd_stream(d, win, win_row, F, None, 0, Y0, 2 * H, T, dirn, hn, dirn * H, 4 * H, st, T)
                row.update({GF["WPX"]: ops.addr(d.wp_x), GF["BGI"]: ops.addr(d.bias_gi), GF["XF"]: F})
            else:
                gi = self.buf(f"gi_e0_{dirn}", B, T, 3 * H)
                ops.gemm(B * T, 3 * H, F, x_op, 0, self.P(d.w_ih, F), 0, gi, 3 * H, bias=d.bias_gi)
                row = self._gru_fwd_stream(d, gi, T * 3 * H, 3 * H, None, 0, Y0, 2 * H, T, dirn, hn, dirn * H, 4 * H, st, T)
            rows.append(row)
        self._gru_fwd(rows, B)
        y_op = Operand(Y0, 2 * H, off=2 * H, seg=T, seg_stride=(T + 2) * 2 * H)
        self._drop_mask = None
        if training and s.dropout > 0:
            if drop_mask is None:
                raise ValueError("encode(training=True) with dropout_encoder > 0 needs a dropout mask")
            Y0d = self.buf("Y0d", B, T, 2 * H)
            ops.mask_scale(Y0, 2 * H, 2 * H, T, (T + 2) * 2 * H, drop_mask, 1.0 / (1.0 - s.dropout), Y0d, B * T, 2 * H)
            y_op = Operand(Y0d, 2 * H)
            self._drop_mask = drop_mask
        Y1 = self.buf("Y1", B, T + 2, 2 * H) if (training or coop) else None
        rows, jobs = [], []
        for dirn, d in enumerate(self.enc[1]):
            gi = self.buf(f"gi_e1_{dirn}", B, T, 3 * H)
            jobs.append(lambda d=d, gi=gi: ops.gemm(B * T, 3 * H, 2 * H, y_op, 0, self.P(d.w_ih, 2 * H), 0, gi, 3 * H, bias=d.bias_gi,
                                                    split=self._split_rows(B * T, 3 * H, 2 * H, y_op, self.P(d.w_ih, 2 * H), 0)))
            st = self.buf(f"st_e1_{dirn}", ops.gru_stash_floats(B, T, H)) if training else None
            rows.append(self._gru_fwd_stream(d, gi, T * 3 * H, 3 * H, None, 0, Y1, 2 * H, T, dirn, hn, (2 + dirn) * H, 4 * H, st, T,
                                             write_y=training or coop))
        # the two directions' input projections fill each other's tails (small batches only: see wgrad_streams)
        self._parallel(jobs, min(2, self.small_streams) if B <= 1024 else 1)
        self._gru_fwd(rows, B)
        return hn

    def loss_sums(self):
        """The 8-float buffer the loss kernels add their sums to (slots LOSS_*).  Zero between steps: allocated zeroed, and
        ops.loss_finish -- the one reader -- leaves it zeroed (so no fill launch opens a step)."""
        return self.ws.get("losses", 8, self.dev, zero=True)

    def seed_rng(self, seed, step=0):
        """(Re)start the device-side N(0,1) stream of the reparameterisation: Philox keyed by `seed`, next draw = `step`."""
        self._rng = torch.tensor([int(seed) & 0x7fffffffffffffff, int(step), 0, 0], dtype=torch.int64, device=self.dev)

    def latent(self, hn, B, eps, training, want_kl=True):
        """eps = None in training mode: the latent kernel draws it (device-side Philox stream, seed_rng) into the engine's own eps buffer."""
        s, H, Z = self.spec, self.spec.H, self.spec.Z
        self.serial += 1
        mu, lvr = self.buf("mu", B, Z), self.buf("lv_raw", B, Z)
        logvar, z = self.buf("logvar", B, Z), self.buf("z", B, Z)
        hn_op = Operand(hn, 4 * H)
        ops.gemm(B, Z, 4 * H, hn_op, 0, self.P("lmbda.hidden_to_mean.weight", 4 * H), 0, mu, Z,
                 bias=self._pv("lmbda.hidden_to_mean.bias"), splitk=0)
        ops.gemm(B, Z, 4 * H, hn_op, 0, self.P("lmbda.hidden_to_logvar.weight", 4 * H), 0, lvr, Z,
                 bias=self._pv("lmbda.hidden_to_logvar.bias"), splitk=0)
        rng = None
        if training and eps is None:
            if self._rng is None:
                self.seed_rng(int(torch.empty((), dtype=torch.int64).random_().item()))       # from torch's (seedable) CPU generator, once
            rng, eps = self._rng, self.buf("eps", B, Z)
        self._eps_used = eps
        ops.latent_fwd(mu, lvr, eps, B, Z, s.softplus, training, logvar, z, self.loss_sums()[LOSS_KLSUM:] if want_kl else None, rng=rng)
        return z, mu, logvar

    def _decode_one(self, tag, name, dirs, steps, z, B, training, rows, jobs, inputs=None):
        H, Z = dirs[0].H, self.spec.Z
        hid = None
        narrow = inputs is None and Z <= 32 and self.group_zproj     # the projections of the time-constant z leave as ONE launch (decode(): ops.linear_group)
        if self.h0_from_z:
            hid = self.buf(f"hid_{tag}", B, 2 * H)
            if narrow:
                jobs.append((self.P(f"{name}.latent_to_hidden.weight", Z), self._pv(f"{name}.latent_to_hidden.bias"), hid, 2 * H, 2 * H))
            else:
                jobs.append(lambda: ops.gemm(B, 2 * H, Z, Operand(z, Z), 0, self.P(f"{name}.latent_to_hidden.weight", Z), 0, hid, 2 * H,
                                             bias=self._pv(f"{name}.latent_to_hidden.bias")))
        Y = self.buf(f"Y_{tag}", B, steps + 2, 2 * H)
        for dirn, d in enumerate(dirs):
            st = self.buf(f"st_{tag}_{dirn}", ops.gru_stash_floats(B, steps, H)) if training else None
            h0_off = dirn * B * H          # hidden.view(2,B,H) (rnn_model.py:104,137): direction d, row b lives at flat offset (d*B+b)*H
            if inputs is None:             # z at every step (rnn_model.py:169-170): one projection, constant in time
                gi = self.buf(f"gi_{tag}_{dirn}", B, 3 * H)
                if narrow:
                    jobs.append((self.P(d.w_ih, Z), d.bias_gi, gi, 3 * H, 3 * H))
                else:
                    jobs.append(lambda d=d, gi=gi: ops.gemm(B, 3 * H, Z, Operand(z, Z), 0, self.P(d.w_ih, Z), 0, gi, 3 * H, bias=d.bias_gi))
                rows.append(self._gru_fwd_stream(d, gi, 3 * H, 0, hid, h0_off, Y, 2 * H, steps, dirn, None, 0, 0, st, steps))
            else:                          # the caller's own sequence (B, L >= steps, Z): a projection per step, like encoder layer 1
                ins, L = inputs
                gi = self.buf(f"gi_seq_{tag}_{dirn}", B, steps, 3 * H)
                jobs.append(lambda d=d, gi=gi: ops.gemm(B * steps, 3 * H, Z, Operand(ins, Z, seg=steps, seg_stride=L * Z), 0,
                                                        self.P(d.w_ih, Z), 0, gi, 3 * H, bias=d.bias_gi))
                rows.append(self._gru_fwd_stream(d, gi, steps * 3 * H, 3 * H, hid, h0_off, Y, 2 * H, steps, dirn, None, 0, 0, st, steps))
        return Y

    def decode(self, z, B, training, which="both", heads=True, inputs=None):
        """Decoder (+ Decoder_Future) from z.  which = "dec" / "fut" runs only that one (the sub-module call patterns
        model.decoder(ins, z) / model.decoder_future(ins, z) of generative_functions.py:39).  inputs = (tensor (B, L, Z) contiguous, L):
        the GRUs run over this sequence instead of z tiled over time (inference only; rnn_model.py:99-109 accepts any `inputs`)."""
        if inputs is not None and training:
            raise ValueError("decode(inputs=...) is an inference path; training builds the decoder input from z (rnn_model.py:169-170)")
        s, H, F, T, FS = self.spec, self.spec.H, self.spec.F, self.spec.T, self.spec.FS
        self.repack()
        self.serial += 1
        rows, jobs = [], []
        want_d, want_f = which in ("both", "dec"), s.future and which in ("both", "fut")
        Yd = self._decode_one("dec", "decoder", self.dec, T, z, B, training, rows, jobs, inputs) if want_d else None
        Yf = self._decode_one("fut", "decoder_future", self.fut, FS, z, B, training, rows, jobs, inputs) if want_f else None
        # <= 6 independent (B x 3H|2H x Z) projections of z: one launch for zdims <= 32 (K <= 32: a store stream, no matrix cores), else GEMMs on streams
        lin = [j for j in jobs if isinstance(j, tuple)]
        for i in range(0, len(lin), 8):
            ops.linear_group(Operand(z, self.spec.Z), B, self.spec.Z, lin[i:i + 8])
        self._parallel([j for j in jobs if not isinstance(j, tuple)], self.small_streams)
        self._gru_fwd(rows, B)
        self._issue_pending_cluster()                 # the solve runs beside the output heads below
        H, Hf = s.Hd, s.Hf
        pred = None
        self._heads_deferred = not heads
        self._dY_ready = False
        if not heads:          # loss() runs the streaming output heads (ops.head_stream): prediction, loss, dpred, dY and dW in one pass over Y
            return self.buf("pred", B, T, F) if want_d else None, self.buf("futp", B, FS, F) if want_f else None
        if want_d:
            pred = self.buf("pred", B, T, F)
            Kd = len(self.dec) * H             # H for the uni-directional legacy decoder: only the first half of each Y row
            ops.gemm(B * T, F, Kd, Operand(Yd, 2 * H, off=2 * H, seg=T, seg_stride=(T + 2) * 2 * H), 0,
                     self.P("decoder.hidden_to_output.weight", Kd), 0, pred, F, bias=self._pv("decoder.hidden_to_output.bias"))
        fut = None
        if want_f:
            fut = self.buf("futp", B, FS, F)
            ops.gemm(B * FS, F, 2 * Hf, Operand(Yf, 2 * Hf, off=2 * Hf, seg=FS, seg_stride=(FS + 2) * 2 * Hf), 0,
                     self.P("decoder_future.hidden_to_output.weight", 2 * Hf), 0, fut, F,
                     bias=self._pv("decoder_future.hidden_to_output.bias"))
        return pred, fut

    def cluster_terms(self, B, kl_weight, kloss, klmbda, bsize):
        """cluster_loss (rnn_vae.py:45-50) from the (Z,Z) Gram: losses[KMEANS] and Minv with d loss/dz = z Minv."""
        Z = self.spec.Z
        z, G = self.buf("z", B, Z), self.buf("gram", Z, Z)
        sk = max(1, min(64, B // 256))
        ws = self.ws.get("splitk_gram", sk * Z * Z, self.dev) if sk > 1 else None
        ops.gemm(Z, Z, B, Operand(z, Z), 1, Operand(z, Z), 1, G, Z, splitk=sk, ws=ws)
        if self._nuc_state is None:
            self._nuc_state = torch.zeros(ops.nuclear_state_doubles(Z), device=self.dev, dtype=torch.float64)
        ops.nuclear(G, Z, kloss, B, klmbda, bsize, self.loss_sums(), LOSS_KMEANS, self.buf("Minv", Z, Z), gscale=kl_weight,
                    vstate=self._nuc_state)

    def set_overlap(self, on):
        """Turn the three side-stream overlaps (nuclear-norm solve, early future-decoder dW_hh, narrow weight gradients) on / off
        together; bench.py's per-kernel pass runs with them off so that a kernel's duration is its own."""
        prev = (self.nuc_side, self.bwd_overlap, self.skinny_side)
        self.nuc_side, self.bwd_overlap, self.skinny_side = (on, on, on) if isinstance(on, bool) else on
        return prev

    def _issue_pending_cluster(self):
        """Launch the deferred nuclear-norm solve on the side stream, ordered after everything issued so far on the caller's."""
        if self._nuc_pending is None:
            return
        args, self._nuc_pending = self._nuc_pending, None
        if self._nuc_stream is None:
            self._nuc_stream = _side_stream(self.dev, "nuc")
        main = torch.cuda.current_stream(self.dev)
        self._nuc_stream.wait_stream(main)
        with torch.cuda.stream(self._nuc_stream):
            self.cluster_terms(*args)
            self._nuc_event = torch.cuda.Event()
            self._nuc_event.record(self._nuc_stream)

    def abandon_step(self):
        """A step raised between its first loss kernel and vame_loss_finish_f32 (the only launch that zeroes the sums): drop a solve that was deferred but not
        issued, wait for one that is running on the side stream (it still adds its term), then zero the sums so that the next step starts clean."""
        self._nuc_pending = None
        self.join_cluster()
        self._join_early()
        self.loss_sums().zero_()

    def join_cluster(self):
        """Before Minv / losses[KMEANS] are read on the caller's stream."""
        if self._nuc_event is not None:
            torch.cuda.current_stream(self.dev).wait_event(self._nuc_event)
            self._nuc_event = None

    def forward(self, win, win_row, B, eps, training, cluster=None, enc_in=None, drop_mask=None, defer_heads=False, want_kl=False):
        """Full RNN_VAE.forward (rnn_model.py:162-179).  Returns workspace views pred, fut, z, mu, logvar.
        cluster = (kl_weight, kloss, klmbda, bsize) also evaluates the nuclear-norm loss as soon as z exists.
        enc_in (B,T,F contiguous) replaces the first T steps of `win` as the encoder input (input-noise option)."""
        s = self.spec
        xin, xin_row = (win, win_row) if enc_in is None else (enc_in, s.T * s.F)
        hn = self.encode(xin, xin_row, B, training, drop_mask=drop_mask)
        # want_kl: only a caller that goes on to loss() + ops.loss_finish (which consumes the sums) asks for the KL sum
        z, mu, logvar = self.latent(hn, B, eps, training, want_kl=want_kl)
        eps = self._eps_used
        if cluster is not None:
            if self.nuc_side and training and self.dev.type == "cuda":
                self._nuc_pending = (B,) + tuple(cluster)             # issued by decode() right behind the decoders' GRU launch
            else:
                self.cluster_terms(B, *cluster)
        self._cluster_done = cluster is not None
        pred, fut = self.decode(z, B, training, heads=not (defer_heads and training and self._heads_fusable(B)))
        self._issue_pending_cluster()                                 # (decode() did it unless it returned early)
        self._B = B
        self._win, self._win_row, self._eps = xin, xin_row, eps
        sh = lambda t, *shape: t[:_numel(shape)].view(*shape)
        return (sh(pred, B, s.T, s.F), sh(fut, B, s.FS, s.F) if fut is not None else None, sh(z, B, s.Z), sh(mu, B, s.Z),
                sh(logvar, B, s.Z))

    # ------------------------------------------------------------------ loss (fused fwd + grad seeds)
    def _heads_fusable(self, B):
        """The streaming output head (vame_head_stream_f32) covers both decoders' Linear layers at this batch (F <= 32, state width a multiple of 64 up to 512)."""
        s = self.spec
        return (self.fuse_heads and ops.head_stream_ws_floats(B * s.T, s.F, len(self.dec) * s.Hd) > 0
                and (not s.future or ops.head_stream_ws_floats(B * s.FS, s.F, 2 * s.Hf) > 0))

    def _head(self, tag, name, dirs, steps, B, tgt, tgt_off, tgt_row, gscale, pred, dpred, losses, slot):
        """One decoder's output head on the training path in one pass over its states: hidden_to_output, MSE(sum), dpred, dY = dpred W and the
        Linear's weight gradient dW = dpred^T Y (straight into the gradient bucket: _decoder_backward then skips both contractions)."""
        H, F = dirs[0].H, self.spec.F
        Ko = len(dirs) * H
        Y = self.buf(f"Y_{tag}", B, steps + 2, 2 * H)
        wo = f"{name}.hidden_to_output.weight"
        ws = self.ws.get(f"head_ws_{tag}", ops.head_stream_ws_floats(B * steps, F, Ko), self.dev)
        ops.head_stream(Operand(Y, 2 * H, off=2 * H, seg=steps, seg_stride=(steps + 2) * 2 * H), B * steps, F, Ko,
                        self.P(wo, Ko), self._pv(f"{name}.hidden_to_output.bias"), tgt, tgt_off, tgt_row, gscale,
                        pred, dpred, self.buf(f"dY_{tag}", B, steps, 2 * H), 2 * H, losses, slot, self.g, self.table.off(wo), ws)

    def loss(self, B, tgt, tgt_row, fut_tgt_off, kl_weight, kloss, klmbda, bsize, mse_red="sum", mse_pred="sum", with_future=True):
        """rnn_vae.py:124-129.  Fills losses[REC,FUT,KLSUM,KMEANS] and the gradient seeds dpred/dfut/Minv."""
        s, T, F, FS, Z = self.spec, self.spec.T, self.spec.F, self.spec.FS, self.spec.Z
        losses = self.loss_sums()
        dpred = self.buf("dpred", B, T, F)
        sc = 2.0 if mse_red == "sum" else 2.0 / (B * T * F)
        if self._heads_deferred:
            self._head("dec", "decoder", self.dec, T, B, tgt, 0, tgt_row, sc, self.buf("pred", B, T, F), dpred, losses, LOSS_REC)
        else:
            ops.mse_fwd_bwd(self.buf("pred", B, T, F), tgt, 0, tgt_row, B, T * F, sc, dpred, losses, LOSS_REC)
        if s.future and with_future:
            dfut = self.buf("dfut", B, FS, F)
            sc = 2.0 if mse_pred == "sum" else 2.0 / (B * FS * F)
            if self._heads_deferred:
                self._head("fut", "decoder_future", self.fut, FS, B, tgt, fut_tgt_off, tgt_row, sc, self.buf("futp", B, FS, F), dfut, losses, LOSS_FUT)
            else:
                ops.mse_fwd_bwd(self.buf("futp", B, FS, F), tgt, fut_tgt_off, tgt_row, B, FS * F, sc, dfut, losses, LOSS_FUT)
        self._dY_ready = self._heads_deferred and (not s.future or with_future)
        if not getattr(self, "_cluster_done", False):
            self.cluster_terms(B, kl_weight, kloss, klmbda, bsize)
        self._cluster_done = False
        return losses

    # ------------------------------------------------------------------ backward
    def _gru_bwd_stream(self, d, stash, Y, y_T, dirn, dY, dy_T, dhn, dhn_off, dhn_row, dG, dh0, dh0_off, dbias, T):
        H = d.H
        spec = SimpleNamespace(d=d, stash=stash, Y=Y, y_T=y_T, dirn=dirn, dY=dY, dy_T=dy_T, dhn=dhn, dhn_off=dhn_off, dhn_row=dhn_row,
                               dG=dG, dh0=dh0, dh0_off=dh0_off, dbias=dbias, T=T)
        return {"_s": spec,
                GB["STASH"]: ops.addr(stash), GB["Y"]: ops.addr(Y, 2 * H + dirn * H), GB["Y_ROW"]: (y_T + 2) * 2 * H,
                GB["Y_T"]: 2 * H, GB["WPT"]: ops.addr(d.wp_bwd),
                GB["DY"]: ops.addr(dY, dirn * H) if dY is not None else 0, GB["DY_ROW"]: dy_T * 2 * H, GB["DY_T"]: 2 * H,
                GB["DHN"]: ops.addr(dhn, dhn_off) if dhn is not None else 0, GB["DHN_ROW"]: dhn_row, GB["DG"]: ops.addr(dG),
                GB["DH0"]: ops.addr(dh0, dh0_off) if dh0 is not None else 0, GB["DH0_ROW"]: H, GB["DBIAS"]: ops.addr(dbias),
                GB["T"]: T, GB["REVERSE"]: dirn, GB["PAD"]: 1}

    def _gru_param_grads(self, d: GruDir, dG, dbias, ntiles, B, T, Yseq, dirn, x_op, x_K, const_in=None):
        """dW_ih, dW_hh, db_ih, db_hh of one (layer,direction) from its dG (B,T,4H) stash."""
        H, g, t = d.H, self.g, self.table
        K = B * T
        dG_i = Operand(dG, 4 * H)                               # [da_r | da_z | dgi_n] = cols 0..3H
        if const_in is None:
            self._gemm_wgrad(3 * H, x_K, K, dG_i, x_op, d.w_ih)
        else:                                                   # time-constant input: sum_t dG first (in-kernel)
            dgsum, z = const_in
            self._gemm_wgrad(3 * H, x_K, B, Operand(dgsum, 3 * H), Operand(z, x_K), d.w_ih)
        # h_{t-1} rows: padded slot t (forward dir) / t+2 (reverse dir) of the (B,T+2,2H) sequence
        hp = Operand(Yseq, 2 * H, off=(2 * 2 * H if dirn else 0) + dirn * H, seg=T, seg_stride=(T + 2) * 2 * H)
        # dW_hh = [da_r | da_z | dgh_n]^T h_{t-1}: columns 0..2H and 3H..4H of dG in one call (skip the dgi_n block)
        self._gemm_wgrad(3 * H, H, K, dG_i, hp, d.w_hh, gap_at=2 * H, gap=H)
        ob_i, ob_h = t.off(d.b_ih), t.off(d.b_hh)
        # bias gradients = column sums of the per-tile partials; queued and reduced in one batched launch per backward
        self._colsum_jobs += [(dbias, 0, ntiles, 3 * H, 4 * H, g, ob_i), (dbias, 0, ntiles, 2 * H, 4 * H, g, ob_h),
                              (dbias, 3 * H, ntiles, H, 4 * H, g, ob_h + 2 * H)]

    def _bias_colsum(self, src, R, C, g_off):
        """Bias gradient = column sum of a dense (R, C) gradient that stays intact until the end of backward().  Small batches: joins the ONE
        batched launch that closes the backward pass (nothing reads flat_g before the optimizer) instead of two launches in the middle of the
        latency-bound chain between the decoders' and the encoder's BPTT launches (batch 256: 8 launches of ~5 us off that chain); large
        batches keep the two-pass reduction (the batched kernel walks a job's rows with four threads per column)."""
        if R <= 1024:
            self._colsum_jobs.append((src, 0, R, C, C, self.g, g_off))
        else:
            ops.colsum(src, 0, R, C, C, self.g, g_off)

    def _decoder_backward(self, tag, name, dirs, steps, dpred, B, dz, first):
        H, F, Z, t = dirs[0].H, self.spec.F, self.spec.Z, self.table
        ntiles = (B + 31) // 32
        Y = self.buf(f"Y_{tag}", B, steps + 2, 2 * H)
        Yrows = Operand(Y, 2 * H, off=2 * H, seg=steps, seg_stride=(steps + 2) * 2 * H)
        dY = self.buf(f"dY_{tag}", B, steps, 2 * H)
        wo = f"{name}.hidden_to_output.weight"
        Ko = len(dirs) * H
        if not getattr(self, "_dY_ready", False):             # (the streaming head of loss() already wrote dY and the Linear's weight gradient)
            ops.gemm(B * steps, Ko, F, Operand(dpred, F), 0, self.P(wo, Ko), 1, dY, 2 * H)
            self._gemm_wgrad(F, Ko, B * steps, Operand(dpred, F), Yrows, wo)
        ops.colsum(dpred, 0, B * steps, F, F, self.g, t.off(f"{name}.hidden_to_output.bias"))
        dhid = self.buf(f"dhid_{tag}", B, 2 * H) if self.h0_from_z else None
        rows, per = [], []
        for dirn, d in enumerate(dirs):
            dG = self.buf(f"dG_{tag}_{dirn}", B, steps, 4 * H)
            dbias = self.buf(f"db_{tag}_{dirn}", ntiles, 4 * H)
            dgsum = self.buf(f"dgs_{tag}_{dirn}", B, 3 * H)
            st = self.buf(f"st_{tag}_{dirn}", ops.gru_stash_floats(B, steps, H))
            rows.append(self._gru_bwd_stream(d, st, Y, steps, dirn, dY, steps, None, 0, 0, dG, dhid, dirn * B * H if dhid is not None else 0,
                                             dbias, steps))
            per.append((d, dG, dbias, dgsum))
        return rows, per, Y, dhid

    def backward(self, B, kl_weight, beta, dz_ext=None, dmu_ext=None, dlv_ext=None, use_minv=True):
        """Backward of the whole model given the seeds left by loss() (or external ones).  Fills flat_g."""
        s, H, F, Z, T, FS, t = self.spec, self.spec.H, self.spec.F, self.spec.Z, self.spec.T, self.spec.FS, self.table
        ntiles = (B + 31) // 32
        self._colsum_jobs = []
        self._wgrad_queue = []
        self._B_bwd = B
        z = self.buf("z", B, Z)
        dz = self.buf("dz", B, Z)
        # ---- decoders (one BPTT launch for all 2 or 4 streams)
        rows_d, per_d, Yd, dhid_d = self._decoder_backward("dec", "decoder", self.dec, T, self.buf("dpred", B, T, F), B, dz, True)
        rows, groups = list(rows_d), [("decoder", per_d, Yd, dhid_d, T)]
        if s.future:
            rows_f, per_f, Yf, dhid_f = self._decoder_backward("fut", "decoder_future", self.fut, FS, self.buf("dfut", B, FS, F), B, dz, False)
            rows += rows_f
            groups.append(("decoder_future", per_f, Yf, dhid_f, FS))
        if self.nuc_join_before_coop and self._nuc_event is not None and any(
                self._coop_parts([r for r in rows if r["_s"].d.H == hh_], B, GB["T"], hh_) for hh_ in {r["_s"].d.H for r in rows}):
            self.join_cluster()                  # (off by default: see nuc_join_before_coop)
        self._gru_bwd(rows, B)
        ev_dec = None
        if self.bwd_overlap and self.dev.type == "cuda" and s.future:
            ev_dec = torch.cuda.Event()
            ev_dec.record()
        He = H
        dz_jobs = []                                                # dz = sum of (B x K) @ (K x Z) products: issued together below
        for name, per, Y, dhid, steps in groups:
            H = per[0][0].H                                          # hidden size of this decoder
            for dirn, (d, dG, dbias, dgsum) in enumerate(per):
                ops.timesum(dG, B, steps, 3 * H, 4 * H, dgsum)       # z is constant in time: sum_t dG first
                self._gru_param_grads(d, dG, dbias, ntiles, B, steps, Y, dirn, None, Z, const_in=(dgsum, z))
                dz_jobs.append((3 * H, Operand(dgsum, 3 * H), self.P(d.w_ih, Z)))
            if dhid is not None:
                wl = f"{name}.latent_to_hidden.weight"
                self._gemm_wgrad(2 * H, Z, B, Operand(dhid, 2 * H), Operand(z, Z), wl)
                self._bias_colsum(dhid, B, 2 * H, t.off(f"{name}.latent_to_hidden.bias"))
                dz_jobs.append((2 * H, Operand(dhid, 2 * H), self.P(wl, Z)))
        if ev_dec is not None:
            Hf_, Kf = s.Hf, B * FS
            # the two dW_hh jobs only (A = dG with the dgi_n block skipped: gap_at = 2H, gap = H).  With FS = 1 and zdims = H the
            # time-constant dW_ih job has the same (M, N, K) -- and reads `dgsum`, which timesum wrote AFTER ev_dec was recorded
            self._early_wgrads(ev_dec, lambda j: j[0] == 3 * Hf_ and j[1] == Hf_ and j[2] == Kf and j[7] == 2 * Hf_ and j[8] == Hf_
                               and j[5].startswith("decoder_future.") and ".weight_hh" in j[5])
        self._sum_into(dz, B, Z, dz_jobs)
        H = He
        self.join_cluster()
        if use_minv and kl_weight != 0:
            ops.gemm(B, Z, Z, Operand(z, Z), 0, Operand(self.buf("Minv", Z, Z), Z), 1, dz, Z, accumulate=True)
        if dz_ext is not None:
            ops.axpy(dz_ext, 1.0, dz, B * Z)
        # ---- reparameterisation + KL
        dmu, dlv = self.buf("dmu", B, Z), self.buf("dlv", B, Z)
        ckl = beta * kl_weight / (B * Z)
        ops.latent_bwd(dz, self.buf("mu", B, Z), self.buf("logvar", B, Z), self.buf("lv_raw", B, Z), self._eps, B, Z,
                       s.softplus, ckl, dmu, dlv)
        if dmu_ext is not None:
            ops.axpy(dmu_ext, 1.0, dmu, B * Z)
        if dlv_ext is not None:
            ops.axpy(dlv_ext, 1.0, dlv, B * Z)
        hn = self.buf("hn", B, 4 * H)
        dhn = self.buf("dhn", B, 4 * H)
        for nm, dv, first in (("lmbda.hidden_to_mean", dmu, True), ("lmbda.hidden_to_logvar", dlv, False)):
            self._gemm_wgrad(Z, 4 * H, B, Operand(dv, Z), Operand(hn, 4 * H), nm + ".weight")
            self._bias_colsum(dv, B, Z, t.off(nm + ".bias"))
            ops.gemm(B, 4 * H, Z, Operand(dv, Z), 0, self.P(nm + ".weight", 4 * H), 1, dhn, 4 * H, accumulate=not first)
        # ---- encoder layer 1
        Y0, Y1 = self.buf("Y0", B, T + 2, 2 * H), self.buf("Y1", B, T + 2, 2 * H)
        rows, per = [], []
        for dirn, d in enumerate(self.enc[1]):
            dG = self.buf(f"dG_e1_{dirn}", B, T, 4 * H)
            dbias = self.buf(f"db_e1_{dirn}", ntiles, 4 * H)
            st = self.buf(f"st_e1_{dirn}", ops.gru_stash_floats(B, T, H))
            rows.append(self._gru_bwd_stream(d, st, Y1, T, dirn, None, T, dhn, (2 + dirn) * H, 4 * H, dG, None, 0, dbias, T))
            per.append((d, dG, dbias))
        self._join_early()                                           # the GRU launch wants every CU
        self._gru_bwd(rows, B)
        dY0 = self.buf("dY0", B, T, 2 * H)
        y0rows = Operand(Y0, 2 * H, off=2 * H, seg=T, seg_stride=(T + 2) * 2 * H)
        if self._drop_mask is not None:             # layer 1 saw the dropped sequence: its dW_ih contracts with that one
            y0rows = Operand(self.buf("Y0d", B, T, 2 * H), 2 * H)
        for dirn, (d, dG, dbias) in enumerate(per):
            ops.gemm(B * T, 2 * H, 3 * H, Operand(dG, 4 * H), 0, self.P(d.w_ih, 2 * H), 1, dY0, 2 * H, accumulate=dirn > 0,
                     split=self._split_rows(B * T, 2 * H, 3 * H, Operand(dG, 4 * H), self.P(d.w_ih, 2 * H), 1))
            self._gru_param_grads(d, dG, dbias, ntiles, B, T, Y1, dirn, y0rows, 2 * H)
        if self._drop_mask is not None:             # d(Y0 * m / (1-p)) / dY0
            ops.mask_scale(dY0, 0, 2 * H, 0, 0, self._drop_mask, 1.0 / (1.0 - s.dropout), dY0, B * T, 2 * H)
        # ---- encoder layer 0
        rows, per = [], []
        for dirn, d in enumerate(self.enc[0]):
            dG = self.buf(f"dG_e0_{dirn}", B, T, 4 * H)
            dbias = self.buf(f"db_e0_{dirn}", ntiles, 4 * H)
            st = self.buf(f"st_e0_{dirn}", ops.gru_stash_floats(B, T, H))
            rows.append(self._gru_bwd_stream(d, st, Y0, T, dirn, dY0, T, dhn, dirn * H, 4 * H, dG, None, 0, dbias, T))
            per.append((d, dG, dbias))
        self._gru_bwd(rows, B)
        xrows = Operand(self._win, F, seg=T, seg_stride=self._win_row)
        for dirn, (d, dG, dbias) in enumerate(per):
            self._gru_param_grads(d, dG, dbias, ntiles, B, T, Y0, dirn, xrows, F)
        self._flush_wgrads()
        ops.colsum_batch(self._colsum_jobs)
        self._colsum_jobs = []


def _slabs(K, sk):
    """k-slabs vame_gemm_f32 / vame_gemm_group_f32 make of K for a requested split-K of sk (slabs are whole 32-row k-tiles)."""
    kper = -(-(-(-K // sk)) // 32) * 32
    return -(-K // kper)


def _numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n
