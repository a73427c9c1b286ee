"""Thin tensor-level wrappers over the C ABI (include/vame_hip.h).  PyTorch is plumbing here:
device memory, the current HIP stream and nothing else."""
import ctypes

import torch

from . import _lib

GF = dict(GI=0, GI_ROW=1, GI_T=2, WP=3, BHN=4, H0=5, H0_ROW=6, Y=7, Y_ROW=8, Y_T=9, HN=10, HN_ROW=11, STASH=12, T=13,
          REVERSE=14, PAD=15, WPX=16, BGI=17, XF=18, OPT=19, N=20)
GB = dict(STASH=0, Y=1, Y_ROW=2, Y_T=3, H0=4, H0_ROW=5, WPT=6, DY=7, DY_ROW=8, DY_T=9, DHN=10, DHN_ROW=11, DG=12, DH0=13,
          DH0_ROW=14, DBIAS=15, OPT=16, T=17, REVERSE=18, PAD=19, N=20)
# enum vame_gru_kernel / VAME_GRU_OPT(...) of include/vame_hip.h: launch options travel in the descriptor table (GF_OPT / GB_OPT of
# stream 0), never through the process environment
KERNEL_AUTO, KERNEL_LOCKSTEP, KERNEL_WS, KERNEL_SKEWED = 0, 1, 2, 3


def gru_opt(kernel=KERNEL_AUTO, pace_cp=-1, pace_ld=-1):
    """The GF_OPT / GB_OPT word: kernel in bits 0-7, pacing values + 1 (0 = default) in bits 8-15 / 16-23."""
    if not (0 <= int(kernel) < 16 and -1 <= int(pace_cp) < 255 and -1 <= int(pace_ld) < 255):      # (which kernels exist is the library's answer)
        raise ValueError(f"gru_opt: kernel={kernel} (a kernel number < 16), pace_cp={pace_cp}, pace_ld={pace_ld} (-1 = default, or 0..254): "
                         "the value would spill into the neighbouring field of the option word")
    return int(kernel) | ((pace_cp + 1 if pace_cp >= 0 else 0) << 8) | ((pace_ld + 1 if pace_ld >= 0 else 0) << 16)


def _stream():
    return _lib.stream_handle()


def _ptr(t, off=0):
    """Device address of element `off` of t.  A train step at the reference's stock batch (256) is ~110 launches with ~500 pointer
    arguments and is HOST-bound (2.5 ms of enqueue against ~2.3 ms of kernels), so this path is kept short: the device check is one
    attribute read, and the hook that refuses host tensors (there is no CPU fallback; the CPU test-suite's harness replaces it) is
    only called when that read says "not on the device"."""
    if t is None:
        return None
    if not t.is_cuda:
        _lib.require_device_tensor(t)
    return t.data_ptr() + off * t.element_size()


def addr(t, off=0):
    """Integer device address of element `off` of tensor t (0 for None) for descriptor tables."""
    return 0 if t is None else _ptr(t, off)


class Operand:
    """A GEMM operand: base tensor (+ element offset), leading dim and optional 2-level row addressing."""
    __slots__ = ("t", "off", "ld", "seg", "seg_stride")

    def __init__(self, t, ld, off=0, seg=0, seg_stride=0):
        self.t, self.off, self.ld, self.seg, self.seg_stride = t, off, ld, seg, seg_stride


_ws_cache = {}
# Bumped whenever a scratch / workspace buffer that launches keep addresses of is (re)allocated -- by this module's caches, by
# engine.Workspace, by CoopState.  A captured step graph holds such addresses: rnn_vae.GraphedTrainStep compares the value it saw at capture
# time and captures again when it moved (e.g. an embedding pass with a larger batch grew the engine's workspace between two epochs).
ALLOC_GEN = [0]


def _auto_ws(dev, n):
    t = _ws_cache.get(dev)
    if t is None or t.numel() < n:
        t = _ws_cache[dev] = torch.empty(max(n, 1 << 20), device=dev)
        ALLOC_GEN[0] += 1
    return t


def gemm(M, N, K, A, a_kmajor, B, b_kmajor, C, ldc, c_off=0, bias=None, accumulate=False, splitk=1, ws=None, a_gap_at=0,
         a_gap=0, split=None):
    """C (+)= op(A) op(B) (+ bias) (vame_gemm_f32).  split = an `opt` word (0 = defaults): the error-compensated split-bf16 form
    (vame_gemm_bf16x6_f32: a row-major A times a plain weight matrix, no split-K; see gemm_split_rows_ok)."""
    L = _lib.lib()
    if split is not None:
        assert not a_kmajor and splitk in (0, 1) and not a_gap and not B.seg, "the split-bf16 form takes a row-major A, a plain B and no split-K"
        rc = L.vame_gemm_bf16x6_f32(M, N, K, _ptr(A.t, A.off), A.ld, A.seg, A.seg_stride, _ptr(B.t, B.off), B.ld, int(b_kmajor), _ptr(bias),
                                    _ptr(C, c_off), ldc, int(accumulate), int(split), _stream())
        _lib.check(rc, "vame_gemm_bf16x6_f32")
        return
    if splitk == 0:
        # auto: few output tiles but a long K (Lambda / latent_to_hidden / dz GEMMs, M = batch): spread K over more workgroups
        tiles = ((M + 127) // 128) * ((N + 127) // 128 if N > 64 else 1)
        splitk = max(1, min(7, 192 // tiles, K // 128)) if tiles <= 64 else 1       # (measured against 16 and never: +1.2 % / +6 % of the batch 4096 / 256 step)
        if splitk > 1:
            ws = _auto_ws(C.device, splitk * M * N)
    if splitk > 1:
        assert ws is not None and ws.numel() >= splitk * M * N, "split-K workspace too small"
    rc = L.vame_gemm_f32(M, N, K, _ptr(A.t, A.off), A.ld, int(a_kmajor), A.seg, A.seg_stride, _ptr(B.t, B.off), B.ld,
                         int(b_kmajor), B.seg, B.seg_stride, _ptr(bias), _ptr(C, c_off), ldc, int(accumulate), splitk,
                         _ptr(ws), a_gap_at, a_gap, _stream())
    _lib.check(rc, "vame_gemm_f32")


def gemm_group(M, N, K, As, a_kmajor, Bs, b_kmajor, C, c_offs, ldc, splitk, ws, accumulate=False, a_gap_at=0, a_gap=0, split=None):
    """len(As) problems of one shape / layout in one launch: C[c_offs[g] ...] (+)= op(As[g]) op(Bs[g]) (vame_gemm_group_f32).
    As / Bs: Operands that differ only in their base (tensor + offset).  split = an `opt` word (0 = defaults): the contraction runs
    as the error-compensated split-bf16 form (vame_gemm_group_bf16x6_f32; both operands k-major, see gemm_split_ok)."""
    n = len(As)
    a0, b0 = As[0], Bs[0]
    assert all((o.ld, o.seg, o.seg_stride) == (a0.ld, a0.seg, a0.seg_stride) for o in As)
    assert all((o.ld, o.seg, o.seg_stride) == (b0.ld, b0.seg, b0.seg_stride) for o in Bs)
    assert ws is not None and ws.numel() >= n * splitk * M * N, "split-K workspace too small"
    arr = ctypes.c_void_p * n
    pa = arr(*[_ptr(o.t, o.off) for o in As])
    pb = arr(*[_ptr(o.t, o.off) for o in Bs])
    pc = arr(*[_ptr(C, off) for off in c_offs])
    if split is not None:
        assert a_kmajor and b_kmajor, "the split-bf16 contraction takes two k-major operands"
        rc = _lib.lib().vame_gemm_group_bf16x6_f32(n, M, N, K, pa, a0.ld, a0.seg, a0.seg_stride, pb, b0.ld, b0.seg, b0.seg_stride, pc, ldc,
                                                   int(accumulate), splitk, _ptr(ws), a_gap_at, a_gap, int(split), _stream())
        _lib.check(rc, "vame_gemm_group_bf16x6_f32")
        return
    rc = _lib.lib().vame_gemm_group_f32(n, M, N, K, pa, a0.ld, int(a_kmajor), a0.seg, a0.seg_stride, pb, b0.ld, int(b_kmajor), b0.seg,
                                        b0.seg_stride, pc, ldc, int(accumulate), splitk, _ptr(ws), a_gap_at, a_gap, _stream())
    _lib.check(rc, "vame_gemm_group_f32")


def linear_group(A, M, K, problems):
    """Up to 8 Linear layers of one narrow input in one launch (vame_linear_group_f32): A = Operand (M, K) with K <= 32; problems =
    [(W Operand (N, K) contiguous, bias tensor or None, C tensor, ldc, N)], C (M, N) with row stride ldc."""
    n = len(problems)
    assert 1 <= n <= 8 and K <= 32 and not A.seg and all(W.ld == K and not W.seg for W, *_ in problems)
    vp, i64, i32 = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int * n
    rc = _lib.lib().vame_linear_group_f32(n, M, K, _ptr(A.t, A.off), A.ld, vp(*[_ptr(W.t, W.off) for W, *_ in problems]),
                                          vp(*[_ptr(b) for _, b, *_ in problems]), vp(*[_ptr(C) for _, _, C, *_ in problems]),
                                          i64(*[p[3] for p in problems]), i32(*[p[4] for p in problems]), _stream())
    _lib.check(rc, "vame_linear_group_f32")


def gemm_split_ok(M, N, K, As, Bs, splitk, a_gap_at=0, a_gap=0):
    """Whether vame_gemm_group_bf16x6_f32 takes this group (its alignment rules; shapes it is worth using for are the caller's choice)."""
    a0, b0 = As[0], Bs[0]
    even = (M, N, a0.ld, b0.ld, a0.seg_stride, b0.seg_stride, a_gap_at, a_gap)
    if any(v % 2 for v in even):
        return False
    if any(_ptr(o.t, o.off) % 8 for o in list(As) + list(Bs)):
        return False
    kper = -(-(-(-K // splitk)) // 32) * 32
    span = max(a0.ld, b0.ld, (a0.seg_stride // a0.seg) if a0.seg else 0, (b0.seg_stride // b0.seg) if b0.seg else 0) * 4 * (kper + 64)
    return -(-K // kper) >= 8 and span < (1 << 30)


def gemm_split_rows_ok(M, N, K, A, a_kmajor, B, b_kmajor):
    """Whether vame_gemm_bf16x6_f32 takes this contraction (its layout / alignment rules; which shapes it pays for is the caller's choice)."""
    if a_kmajor or B.seg or K % 32 or A.ld % 4 or A.seg_stride % 4 or _ptr(A.t, A.off) % 16 or A.ld < K:
        return False
    wrap = (A.seg_stride - A.seg * A.ld) if A.seg else 0
    if wrap < 0 or (128 * A.ld + ((128 // A.seg + 2) if A.seg else 0) * wrap + K) * 4 >= (1 << 30):
        return False
    if b_kmajor:
        return not (B.ld % 2 or N % 2 or _ptr(B.t, B.off) % 8 or B.ld < N or K * B.ld * 4 >= (1 << 30))
    return not (B.ld % 4 or _ptr(B.t, B.off) % 16 or B.ld < K or 128 * B.ld * 4 >= (1 << 30))


class ClockProbe:
    """Average shader clock over a stretch of the current stream (vame_clock_stamp before and after it): `start()`, work, `stop()`,
    then `mhz()` once the stream has been synchronised.  Measurement only (bench.py: roofline.clock_mhz / frac_at_clock)."""
    NB = 256                                                 # workgroups per stamp: a few dozen per XCD

    def __init__(self, dev):
        self.buf = torch.zeros(2, self.NB, 4, dtype=torch.int64, device=dev)

    def _stamp(self, k):
        rc = _lib.lib().vame_clock_stamp(self.buf[k].data_ptr(), self.NB, _stream())
        _lib.check(rc, "vame_clock_stamp")

    def start(self):
        self._stamp(0)

    def stop(self):
        self._stamp(1)

    def mhz(self):
        """Median over compute units of d(shader ticks) / d(100 MHz ticks) x 100; None when the counters did not move (host emulator).
        Only stamps taken on the SAME compute unit (XCC id + the SE / SH / CU bits of HW_ID) are paired: s_memtime counters of
        different CUs are offset against each other by millions of ticks (measured, tools/clock_check.py), which is harmless over a
        460 ms region and garbage over a 30 us kernel."""
        b = self.buf.cpu().numpy()
        first = {}
        for t, r, xcc, hw in b[0]:
            first.setdefault((int(xcc), int(hw) & 0xFF00), (int(t), int(r)))
        vals, seen = [], set()
        for t, r, xcc, hw in b[1]:
            key = (int(xcc), int(hw) & 0xFF00)
            if key in first and key not in seen:
                seen.add(key)
                dt, dr = int(t) - first[key][0], int(r) - first[key][1]
                if dr > 0 and dt > 0:
                    vals.append(100.0 * dt / dr)
        vals.sort()
        return vals[len(vals) // 2] if vals else None


def window_gather(X, N, F, starts, start0, B, L, out):
    rc = _lib.lib().vame_window_gather_f32(_ptr(X), N, F, _ptr(starts), start0, B, L, _ptr(out), _stream())
    _lib.check(rc, "vame_window_gather_f32")


def gru_pack(W_hh, b_ih, b_hh, H, wp_fwd, wp_bwd, bias_gi, b_hn):
    rc = _lib.lib().vame_gru_pack_f32(_ptr(W_hh), _ptr(b_ih), _ptr(b_hh), H, _ptr(wp_fwd), _ptr(wp_bwd), _ptr(bias_gi),
                                      _ptr(b_hn), _stream())
    _lib.check(rc, "vame_gru_pack_f32")


GP = dict(W_HH=0, B_IH=1, B_HH=2, H=3, WP_FWD=4, WP_BWD=5, BIAS_GI=6, B_HN=7, W_IH=8, F=9, WPX=10, N=11)     # enum vame_gru_pack_field
GRU_PACK_MAX = 16


def gru_pack_batch(items):
    """items: [(W_hh, b_ih, b_hh, H, wp_fwd, wp_bwd, bias_gi, b_hn, W_ih | None, F, wpx | None)] -- all of them in one launch."""
    for i in range(0, len(items), GRU_PACK_MAX):
        part = items[i:i + GRU_PACK_MAX]
        tab = (ctypes.c_int64 * (len(part) * GP["N"]))()
        for k, (W_hh, b_ih, b_hh, H, wpf, wpb, bgi, bhn, W_ih, F, wpx) in enumerate(part):
            row = [_ptr(W_hh), _ptr(b_ih), _ptr(b_hh), H, _ptr(wpf), _ptr(wpb), _ptr(bgi), _ptr(bhn), _ptr(W_ih) if W_ih is not None else 0,
                   F, _ptr(wpx) if wpx is not None else 0]
            for f, v in enumerate(row):
                tab[k * GP["N"] + f] = int(v or 0)
        rc = _lib.lib().vame_gru_pack_batch_f32(ctypes.addressof(tab), len(part), _stream())
        _lib.check(rc, "vame_gru_pack_batch_f32")


def gru_pack_x(W_ih, F, H, wpx):
    rc = _lib.lib().vame_gru_pack_x_f32(_ptr(W_ih), F, H, _ptr(wpx), _stream())
    _lib.check(rc, "vame_gru_pack_x_f32")


def gru_stash_floats(B, T, H):
    return int(_lib.lib().vame_gru_stash_floats(B, T, H))


class _Desc:
    """Host-side int64 descriptor table of a GRU launch (nstreams x nfields), filled from the stream dicts.  A ctypes array, not a
    tensor: element-wise tensor indexing cost ~0.3 ms of host time per launch, six launches per step."""
    __slots__ = ("arr", "n")

    def __init__(self, rows, nfields):
        self.n = nfields
        self.arr = (ctypes.c_int64 * (len(rows) * nfields))()
        a = self.arr
        for i, r in enumerate(rows):
            base = i * nfields
            for k, v in r.items():
                if k.__class__ is int:
                    a[base + k] = int(v)

    def __setitem__(self, key, value):
        self.arr[key[0] * self.n + key[1]] = int(value)

    def data_ptr(self):
        return ctypes.addressof(self.arr)


def _desc_tensor(rows, nfields):
    return _Desc(rows, nfields)


def gru_seq_fwd_has_kernel(H, kernel):
    return bool(_lib.lib().vame_gru_seq_fwd_has_kernel(int(H), int(kernel)))


def gru_seq_fwd(streams, B, H, kernel=KERNEL_AUTO, prio=-1, delay=-1):
    """streams: list of dicts keyed by GF[...] indices.  kernel: KERNEL_AUTO / KERNEL_LOCKSTEP / KERNEL_SKEWED (an argument of the call:
    GF_OPT of stream 0); prio / delay: priority mode and part-0 start delay (x 256 cycles) of the skewed kernel (-1 = default)."""
    d = _desc_tensor(streams, GF["N"])
    d[0, GF["OPT"]] = gru_opt(kernel, prio, delay)
    rc = _lib.lib().vame_gru_seq_fwd_f32(d.data_ptr(), len(streams), B, H, _stream())
    _lib.check(rc, "vame_gru_seq_fwd_f32")


class CoopState:
    """Flag words + launch epoch + poll-timeout counter shared by the cooperative (column-split) GRU launches of one device.

    Failure handling: a launch that gives up waiting for a group member increments `status` on the device.  The optimizer
    kernel reads the same word and drops the step (vame_adam_amsgrad_f32 abort_flag), so undefined gradients never reach
    the weights; the host learns about it from `poll()` -- an asynchronous 4-byte copy into pinned memory after every step,
    looked at when the next step is enqueued (no stall) -- or from `check()` wherever it synchronises anyway.

    Several ranks: `shared` is the float word behind the gradient bucket that travels through the SAME all-reduce (SUM) as the
    gradients (rnn_vae.allreduce_gradients writes `status` into it first).  It is then the optimizer's abort word and the word the
    snapshots copy, so one rank's failure drops the step on EVERY rank and every rank raises -- replicas never diverge and nobody
    is left waiting in the next collective.  `on_error` callbacks run (host synchronised, status already cleared) just before the
    exception: the optimizer takes the dropped launches out of its bias-correction step count there."""

    def __init__(self, dev, ints=1 << 16):
        self.flags = torch.full((ints,), -7, dtype=torch.int32, device=dev)      # older than the first epoch, equal to no tag the first launches look for
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        # {launch epoch, ticket}: read and advanced by the kernels themselves (vame_gru_coop_fwd_f32), so that a captured graph can replay them
        self.epoch = torch.tensor([1, 0], dtype=torch.int32, device=dev)
        self.shared = None
        self.on_error = []
        self.dirty = False                                   # cooperative launches since the last check()
        self._host = [torch.zeros(1, dtype=torch.int32, pin_memory=dev.type == "cuda") for _ in range(2)]
        self._ev = [None, None]
        self._k = 0

    def ensure_flags(self, ints):
        """Flag words + the forward kernel's tagged hand-off packets (vame_gru_coop_flag_ints).  A larger buffer starts out as if a launch
        a few epochs back had left it behind: older than anything the next launch waits for, equal to no tag it will look for.  (Reads
        the device's epoch: a host sync, but only when the buffer grows -- the first launches of a shape.)"""
        if self.flags.numel() < ints:
            stale = (int(self.epoch[0].item()) - 8) & 0xffffffff
            self.flags = torch.full((int(ints),), stale - (1 << 32) if stale >= (1 << 31) else stale, dtype=torch.int32, device=self.flags.device)
            ALLOC_GEN[0] += 1

    def snapshot(self):
        """Enqueue a copy of the status word to the host (end of a step); never blocks."""
        if not self.dirty:
            return
        k = self._k = self._k ^ 1
        if self._ev[k] is not None:
            self._ev[k].synchronize()                        # the copy that used this buffer two steps ago (long finished)
            self._raise_if(int(self._host[k][0]))
        self._host[k].copy_(self.status if self.shared is None else self.shared, non_blocking=True)
        if self.status.is_cuda:
            self._ev[k] = torch.cuda.Event()
            self._ev[k].record()
        else:
            self._raise_if(int(self._host[k][0]))

    def poll(self):
        """Look at the snapshots that have already arrived (start of a step); never blocks.  Not with several ranks: WHEN a copy
        arrives differs between ranks, and a rank that raised here would leave the others waiting in the next all-reduce -- there
        every rank examines the same (all-reduced) word at the same program point, in snapshot() two steps later or in check()."""
        if self.shared is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            return
        for k in (0, 1):
            ev = self._ev[k]
            if ev is not None and ev.query():
                self._ev[k] = None
                self._raise_if(int(self._host[k][0]))

    def _raise_if(self, n):
        if n:
            self.status.zero_()
            if self.shared is not None:
                self.shared.zero_()
            self._ev = [None, None]
            self.dirty = False
            for cb in self.on_error:
                cb()
            raise _lib.VameHipError(f"cooperative GRU kernel: {n} hand-off wait(s) timed out (workgroups of a group were not co-resident); "
                                    "the affected optimizer step was dropped on the device; pass engine_options={'coop': False} "
                                    "(config.yaml: `vame_amd_engine: {coop: false}`) to use the batch-tile-persistent kernels")

    def check(self, reduce=None):
        """Host sync: raise if a cooperative launch ever gave up waiting for a group member (its results were undefined).
        reduce: several ranks at the same program point -- maps this rank's count to the maximum over ranks (a collective, so it runs
        whether or not this rank launched anything since the last check: every rank must enter it)."""
        if not self.dirty and reduce is None:
            return
        n = int(self.status.item()) if self.dirty else 0
        if self.shared is not None and self.dirty:
            n = max(n, int(self.shared.item() != 0))
        self.dirty = False
        if reduce is not None:
            n = reduce(n)
        self._raise_if(n)


def gru_coop_set_poll_limit(polls):
    """Poll budget of the cooperative kernels' hand-off waits (0 = default); returns the previous value."""
    return int(_lib.lib().vame_gru_coop_set_poll_limit(int(polls)))


def gru_coop_supported(nstreams, B, H):
    return bool(_lib.lib().vame_gru_coop_supported(nstreams, B, H))


def coop_row_chunks(nstreams, B, H, max_rounds=2):
    """Row ranges [(row0, nrows)] that each fit one cooperative launch, or [] when that would take more than `max_rounds`
    launches (then the batch-tile-persistent kernels are the better choice)."""
    if H not in (128, 256):
        return []
    cap = (256 // (H // 32) // 8 * 8) // nstreams * 32           # rows per launch: groups are dealt to XCDs in eights
    if cap <= 0 or not gru_coop_supported(nstreams, min(cap, B), H):   # (also checks the device's CU count)
        return []
    n = -(-B // cap)
    return [(r, min(cap, B - r)) for r in range(0, B, cap)] if n <= max_rounds else []


def gru_coop_fwd(streams, B, H, state: CoopState, rows=(0, 0), kernel=KERNEL_AUTO):
    """Column-split forward for small batches: same `streams` table as gru_seq_fwd, results equal to summation-order rounding (the
    K = H contraction is summed in two halves).  rows = (row0, nrows) restricts the launch to a row range of the batch (multiple of
    32; (0, 0) = everything).  kernel: KERNEL_AUTO = 16-row groups where twice the workgroups fit one per CU, KERNEL_LOCKSTEP = 32-row
    groups always (the same bits either way)."""
    d = _desc_tensor(streams, GF["N"])
    d[0, GF["OPT"]] = gru_opt(kernel, -1, -1)
    state.ensure_flags(_lib.lib().vame_gru_coop_flag_ints(len(streams), rows[1] or B, H))
    state.dirty = True
    rc = _lib.lib().vame_gru_coop_fwd_f32(d.data_ptr(), len(streams), B, H, rows[0], rows[1], _ptr(state.flags), state.flags.numel(),
                                          _ptr(state.epoch), _ptr(state.status), _stream())
    _lib.check(rc, "vame_gru_coop_fwd_f32")


def gru_coop_bwd(streams, B, H, state: CoopState, rows=(0, 0), kernel=KERNEL_AUTO):
    d = _desc_tensor(streams, GB["N"])
    d[0, GB["OPT"]] = gru_opt(kernel, -1, -1)
    need = _lib.lib().vame_gru_coop_xbuf_floats(len(streams), rows[1] or B, H)
    if getattr(state, "xbuf", None) is None or state.xbuf.numel() < need:
        state.xbuf = torch.empty(need, device=state.flags.device)
        ALLOC_GEN[0] += 1
    state.dirty = True
    rc = _lib.lib().vame_gru_coop_bwd_f32(d.data_ptr(), len(streams), B, H, rows[0], rows[1], _ptr(state.xbuf), _ptr(state.flags), state.flags.numel(),
                                          _ptr(state.epoch), _ptr(state.status), _stream())
    _lib.check(rc, "vame_gru_coop_bwd_f32")


def gru_seq_bwd_has_kernel(H, kernel):
    return bool(_lib.lib().vame_gru_seq_bwd_has_kernel(int(H), int(kernel)))


def gru_seq_bwd(streams, B, H, kernel=KERNEL_AUTO, pace_cp=-1, pace_ld=-1):
    """kernel: KERNEL_AUTO (measured default per hidden size) / KERNEL_LOCKSTEP / KERNEL_WS (wave-specialised; raises where it is not
    instantiated); pace_*: pacing of the wave-specialised kernel's memory waves (-1 = default).  All of it travels in GB_OPT."""
    d = _desc_tensor(streams, GB["N"])
    d[0, GB["OPT"]] = gru_opt(kernel, pace_cp, pace_ld)
    rc = _lib.lib().vame_gru_seq_bwd_f32(d.data_ptr(), len(streams), B, H, _stream())
    _lib.check(rc, "vame_gru_seq_bwd_f32")


def gru_wide_supported(H):
    return bool(_lib.lib().vame_gru_wide_supported(H))


def gru_wide_fwd(streams, B, H, kernel=KERNEL_AUTO):
    """Persistent forward for 256 < H <= 512 (gru_wide.hip); same `streams` table as gru_seq_fwd.  kernel: KERNEL_AUTO / KERNEL_LOCKSTEP /
    KERNEL_SKEWED (H a multiple of 128)."""
    d = _desc_tensor(streams, GF["N"])
    d[0, GF["OPT"]] = gru_opt(kernel)
    rc = _lib.lib().vame_gru_wide_fwd_f32(d.data_ptr(), len(streams), B, H, _stream())
    _lib.check(rc, "vame_gru_wide_fwd_f32")


def gru_wide_bwd(streams, B, H):
    d = _desc_tensor(streams, GB["N"])
    rc = _lib.lib().vame_gru_wide_bwd_f32(d.data_ptr(), len(streams), B, H, _stream())
    _lib.check(rc, "vame_gru_wide_bwd_f32")


def gru_cell_bwd_frag(stash, T, t, dh, dy, dy_off, dy_row, dG, dg_off, dg_row, dgh, B, H):
    rc = _lib.lib().vame_gru_cell_bwd_frag_f32(_ptr(stash), T, t, _ptr(dh), _ptr(dy, dy_off), dy_row, _ptr(dG, dg_off), dg_row, _ptr(dgh), B, H, _stream())
    _lib.check(rc, "vame_gru_cell_bwd_frag_f32")


def gru_cell_fwd(gi, gi_off, gi_row, gh, bhn, hprev, hp_off, hp_row, hout, ho_off, ho_row, stash, st_off, st_row, B, H):
    rc = _lib.lib().vame_gru_cell_fwd_f32(_ptr(gi, gi_off), gi_row, _ptr(gh), _ptr(bhn), _ptr(hprev, hp_off), hp_row,
                                          _ptr(hout, ho_off), ho_row, _ptr(stash, st_off), st_row, B, H, _stream())
    _lib.check(rc, "vame_gru_cell_fwd_f32")


def gru_cell_bwd(stash, st_off, st_row, dh, dy, dy_off, dy_row, dG, dg_off, dg_row, dgh, B, H):
    rc = _lib.lib().vame_gru_cell_bwd_f32(_ptr(stash, st_off), st_row, _ptr(dh), _ptr(dy, dy_off), dy_row, _ptr(dG, dg_off), dg_row,
                                          _ptr(dgh), B, H, _stream())
    _lib.check(rc, "vame_gru_cell_bwd_f32")


def latent_fwd(mu, lv_raw, eps, B, Z, softplus, training, logvar, z, kl_out, rng=None):
    """rng (int64[4] device tensor {seed, step, 0, 0}) in training mode: eps is the OUTPUT of the kernel's own N(0,1) draw and `step` advances on
    the device; rng = None: eps is an input."""
    rc = _lib.lib().vame_latent_fwd_f32(_ptr(mu), _ptr(lv_raw), _ptr(eps), B, Z, int(softplus), int(training), _ptr(logvar),
                                        _ptr(z), _ptr(kl_out), _ptr(rng), _stream())
    _lib.check(rc, "vame_latent_fwd_f32")


_F4 = ctypes.c_float * 4


def loss_finish(raw, scale, weights, with_fut, out, acc=None):
    """raw (8 floats, consumed: zeroed) -> out[0:4] = terms x scale, out[4] = sum weights x terms; acc (float64[6]) += ... (vame_loss_finish_f32)."""
    rc = _lib.lib().vame_loss_finish_f32(_ptr(raw), _F4(*scale), _F4(*weights), int(bool(with_fut)), _ptr(out), _ptr(acc), _stream())
    _lib.check(rc, "vame_loss_finish_f32")


def latent_bwd(dz, mu, logvar, lv_raw, eps, B, Z, softplus, ckl, dmu, dlv):
    rc = _lib.lib().vame_latent_bwd_f32(_ptr(dz), _ptr(mu), _ptr(logvar), _ptr(lv_raw), _ptr(eps), B, Z, int(softplus),
                                        float(ckl), _ptr(dmu), _ptr(dlv), _stream())
    _lib.check(rc, "vame_latent_bwd_f32")


def mse_fwd_bwd(pred, target, tgt_off, tgt_row, B, TF, gscale, dpred, loss_out, loss_off=0):
    rc = _lib.lib().vame_mse_fwd_bwd_f32(_ptr(pred), _ptr(target, tgt_off), tgt_row, B, TF, float(gscale), _ptr(dpred),
                                         _ptr(loss_out, loss_off), _stream())
    _lib.check(rc, "vame_mse_fwd_bwd_f32")


def head_stream_ws_floats(M, F, K):
    """Scratch floats the streaming output head needs for M rows (per-workgroup dW sums), or -1 when it does not cover the shape
    (F > 32, K not a multiple of 64 or above 512): the caller then keeps the separate launches."""
    return int(_lib.lib().vame_head_stream_ws_floats(int(M), int(F), int(K)))


def head_stream(Y, M, F, K, W, bias, tgt, tgt_off, tgt_row, gscale, pred, dpred, dY, dy_ld, loss_out, loss_off, dW, dW_off, ws):
    """Y: Operand over the decoder states (two-level rows (b,t): seg = T, seg_stride); W: Operand (F x K); dW: (F, K) at float offset dW_off of
    a gradient bucket; see vame_head_stream_f32."""
    rc = _lib.lib().vame_head_stream_f32(_ptr(Y.t, Y.off), Y.ld, Y.seg, Y.seg_stride, M, F, K, _ptr(W.t, W.off), _ptr(bias), _ptr(tgt),
                                         tgt_row, tgt_off, float(gscale), _ptr(pred), _ptr(dpred), _ptr(dY), dy_ld, _ptr(loss_out, loss_off),
                                         _ptr(dW, dW_off), _ptr(ws), _stream())
    _lib.check(rc, "vame_head_stream_f32")


def nuclear_state_doubles(Z):
    return int(_lib.lib().vame_nuclear_state_doubles(int(Z)))


def nuclear(G, Z, kloss, nrows, lmbda, bsize, loss_out, loss_off, Minv, gscale=1.0, vstate=None):
    need = nuclear_state_doubles(Z)
    if vstate is None and Z > 64:
        vstate = torch.zeros(need, device=G.device, dtype=torch.float64)         # cold start: no state carried between calls
    if vstate is not None:
        assert vstate.dtype == torch.float64 and vstate.numel() >= need
    rc = _lib.lib().vame_nuclear_f32(_ptr(G), Z, kloss, nrows, float(lmbda), float(bsize), float(gscale),
                                     _ptr(loss_out, loss_off), _ptr(Minv), vstate.data_ptr() if vstate is not None else None,
                                     _stream())
    _lib.check(rc, "vame_nuclear_f32")


def kmeans_assign(X, N, D, C, K, labels, mind2=None, onehot=None, Kp=0):
    assert labels.dtype == torch.int32
    rc = _lib.lib().vame_kmeans_assign_f32(_ptr(X), N, D, _ptr(C), K, labels.data_ptr(), _ptr(mind2), _ptr(onehot), Kp, _stream())
    _lib.check(rc, "vame_kmeans_assign_f32")


def timesum(inp, B, T, C, ld, out):
    rc = _lib.lib().vame_timesum_f32(_ptr(inp), B, T, C, ld, _ptr(out), _stream())
    _lib.check(rc, "vame_timesum_f32")


_colsum_ws = {}


def colsum(inp, in_off, R, C, ld, out, out_off=0, accumulate=False):
    need = int(_lib.lib().vame_colsum_ws_floats(R, C))
    ws = _colsum_ws.get(inp.device)
    if ws is None or ws.numel() < need:
        ws = _colsum_ws[inp.device] = torch.empty(max(need, 1 << 16), device=inp.device)
        ALLOC_GEN[0] += 1
    rc = _lib.lib().vame_colsum_f32(_ptr(inp, in_off), R, C, ld, _ptr(out, out_off), int(accumulate), _ptr(ws), _stream())
    _lib.check(rc, "vame_colsum_f32")


def colsum_batch(jobs):
    """jobs: list of (in_tensor, in_off, R, C, ld, out_tensor, out_off); up to 32 per launch."""
    for i in range(0, len(jobs), 32):
        chunk = jobs[i:i + 32]
        d = torch.tensor([[_ptr(a, ao), R, C, ld, _ptr(o, oo)] for (a, ao, R, C, ld, o, oo) in chunk], dtype=torch.int64)
        rc = _lib.lib().vame_colsum_batch_f32(d.data_ptr(), len(chunk), _stream())
        _lib.check(rc, "vame_colsum_batch_f32")


def adam_amsgrad(p, g, m, v, vmax, n, lr, step, gscale=1.0, beta1=0.9, beta2=0.999, eps=1e-8, abort_flag=None, dropped=None, state=None):
    """state (int32[4] device tensor {lr as float bits, updates applied, 0, 0}): lr / step come from the device and the launch counts itself."""
    rc = _lib.lib().vame_adam_amsgrad_f32(_ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(vmax), n, lr, beta1, beta2, eps, step,
                                          gscale, _ptr(abort_flag), _ptr(dropped), _ptr(state), _stream())
    _lib.check(rc, "vame_adam_amsgrad_f32")


def mask_scale(x, off, ld, seg, seg_stride, mask, scale, out, R, C):
    """out (R,C) = x[two-level rows, see vame_mask_scale_f32] * mask (R,C) * scale."""
    rc = _lib.lib().vame_mask_scale_f32(_ptr(x), off, ld, seg, seg_stride, _ptr(mask), float(scale), _ptr(out), R, C, _stream())
    _lib.check(rc, "vame_mask_scale_f32")


def index_copy(dst, dst_idx, src, src_idx):
    """dst[dst_idx[i]] = src[src_idx[i]] (int64 element indices)."""
    rc = _lib.lib().vame_index_copy_f32(_ptr(dst), _ptr(dst_idx), _ptr(src), _ptr(src_idx), dst_idx.numel(), _stream())
    _lib.check(rc, "vame_index_copy_f32")


def axpy(x, a, y, n, x_off=0, y_off=0):
    rc = _lib.lib().vame_axpy_f32(_ptr(x, x_off), float(a), _ptr(y, y_off), n, _stream())
    _lib.check(rc, "vame_axpy_f32")


# ------------------------------------------------------------------ training-set preparation (float64, (F, N) feature-major)
def prep_zscore_mask(x, F, N, ldx, mean, sd, cutoff, robust, z, ldz, z_off=0):
    rc = _lib.lib().vame_prep_zscore_mask_f64(_ptr(x), F, N, ldx, float(mean), float(sd), float(cutoff), int(bool(robust)),
                                              _ptr(z, z_off), ldz, _stream())
    _lib.check(rc, "vame_prep_zscore_mask_f64")


def prep_ws(F, dev):
    return torch.empty(_lib.lib().vame_prep_ws_bytes(F) // 8, dtype=torch.float64, device=dev)


def prep_fill_last_valid(z, F, N, ld, first_last, ws, z_off=0):
    rc = _lib.lib().vame_prep_fill_last_valid_f64(_ptr(z, z_off), F, N, ld, _ptr(first_last), _ptr(ws), _stream())
    _lib.check(rc, "vame_prep_fill_last_valid_f64")


def prep_fill_across_features(z, F, N, ld, n_empty, z_off=0):
    rc = _lib.lib().vame_prep_fill_across_features_f64(_ptr(z, z_off), F, N, ld, _ptr(n_empty), _stream())
    _lib.check(rc, "vame_prep_fill_across_features_f64")


def prep_rowstats(x, F, N, ld, mean_out, std_out, ws):
    rc = _lib.lib().vame_prep_rowstats_f64(_ptr(x), F, N, ld, _ptr(mean_out), _ptr(std_out), _ptr(ws), _stream())
    _lib.check(rc, "vame_prep_rowstats_f64")


def prep_savgol(x, F, N, ldx, w, L, y, ldy):
    rc = _lib.lib().vame_prep_savgol_f64(_ptr(x), F, N, ldx, _ptr(w), L, _ptr(y), ldy, _stream())
    _lib.check(rc, "vame_prep_savgol_f64")
