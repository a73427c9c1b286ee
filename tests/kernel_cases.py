"""Shared kernel test bodies: run on `cpu` tensors through the emulator or on `cuda` through libvame_hip.so."""
import numpy as np
import pytest
import torch

from oracle import vame_oracle as vo
from vame_amd import ops
from vame_amd.ops import GB, GF, Operand


def T_(a, dev):
    return torch.from_numpy(np.array(a, copy=True, order="C")).to(dev)


def N_(t):
    return t.detach().cpu().numpy()


def check_gemm_group(dev, small=True):
    """vame_gemm_group_f32: several TN / NN problems of one shape in one launch = the single-problem GEMM on each of them, bit
    for bit when the split-K factor is the same (same per-tile k order, same partial-sum order), incl. the column gap of the
    dW_hh form and accumulation into C."""
    rng = np.random.default_rng(3)
    for (M, Nn, K, akm, bkm, sk, n, gap, acc) in ([(96, 136, 512, 1, 1, 8, 3, 0, False), (70, 72, 512, 0, 1, 8, 2, 0, True), (64, 100, 640, 1, 1, 16, 5, 32, False), (200, 30, 512, 1, 1, 8, 4, 0, False), (96, 48, 512, 1, 1, 8, 2, 0, True)]
                                                    + ([] if small else [(768, 256, 8192, 1, 1, 32, 6, 256, False)])):
        Mw = M + (gap or 0)                                    # stored width of a k-major A with a skipped column block
        As = [rng.standard_normal((K, Mw) if akm else (M, K)).astype(np.float32) for _ in range(n)]
        Bs = [rng.standard_normal((K, Nn) if bkm else (Nn, K)).astype(np.float32) for _ in range(n)]
        C0 = rng.standard_normal((n, M, Nn)).astype(np.float32)
        At, Bt = [T_(a, dev) for a in As], [T_(b, dev) for b in Bs]
        Cg, Cs = T_(C0, dev), T_(C0, dev)
        ws = torch.zeros(n * sk * M * Nn, device=dev)
        gap_at = (M // 2) // 4 * 4 if gap else 0
        ops.gemm_group(M, Nn, K, [Operand(a, a.shape[1]) for a in At], akm, [Operand(b, b.shape[1]) for b in Bt], bkm, Cg,
                       [g * M * Nn for g in range(n)], Nn, sk, ws, accumulate=acc, a_gap_at=gap_at, a_gap=gap)
        for g in range(n):
            ops.gemm(M, Nn, K, Operand(At[g], At[g].shape[1]), akm, Operand(Bt[g], Bt[g].shape[1]), bkm, Cs, Nn, c_off=g * M * Nn,
                     accumulate=acc, splitk=sk, ws=ws, a_gap_at=gap_at, a_gap=gap)
        np.testing.assert_array_equal(N_(Cg), N_(Cs), err_msg=str((M, Nn, K, akm, bkm, sk, n, gap)))
        a0 = As[0]
        if akm:
            cols = np.r_[0:gap_at, gap_at + gap:Mw] if gap else np.arange(M)
            a0 = a0[:, cols].T
        ref = a0.astype(np.float64) @ (Bs[0] if bkm else Bs[0].T).astype(np.float64) + (C0[0] if acc else 0)
        np.testing.assert_allclose(N_(Cg)[0], ref, atol=2e-4 * np.sqrt(K))


def split_planes(x):
    """The three bf16 planes of fp32 values by truncation (x = x1 + x2 + x3 exactly), as float64 -- what gemm_split_kernel's producer
    waves compute (vame_amd/csrc/gemm.hip: prod_split)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    hi = lambda v: (v.view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)        # noqa: E731
    x1 = hi(x)
    r1 = x - x1
    x2 = hi(r1)
    x3 = r1 - x2
    assert np.array_equal(x1.astype(np.float64) + x2.astype(np.float64) + x3.astype(np.float64), x.astype(np.float64))
    assert np.array_equal(hi(x3), x3)
    return [v.astype(np.float64) for v in (x1, x2, x3)]


def split_reference(a, b):
    """sum over the six plane products (1,1) (1,2) (2,1) (2,2) (1,3) (3,1) of a^T b in float64: what the split contraction evaluates
    (up to fp32 accumulation rounding)."""
    pa, pb = split_planes(a), split_planes(b)
    return sum(pa[i].T @ pb[j] for i, j in ((0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)))


def check_gemm_split(dev, small=True):
    """vame_gemm_group_bf16x6_f32 (error-compensated split-bf16 contraction of two k-major operands) against float64: ragged and partial
    tiles, partial last k-tile, two-level (batch, time) rows with short and long segments, the dW_hh column gap, accumulation, a shared
    output, both accumulator options.  Tolerances: against the six-plane-product sum the only error is fp32 accumulation (tight);
    against the exact product additionally the dropped plane products (< 2^-21 |a||b| each)."""
    rng = np.random.default_rng(11)
    cases = [  # M, N, K(rows), n, sk, seg_a, seg_b, gap, acc, opt (0 = two accumulators per output, 1 = one)
        (96, 136, 8 * 32, 3, 8, 0, 0, 0, False, 0), (70, 72, 8 * 64 - 2, 2, 8, 0, 6, 0, True, 1), (64, 100, 16 * 30, 2, 8, 30, 30, 32, False, 0),
        (130, 30, 16 * 15, 2, 8, 15, 15, 0, False, 1), (128, 128, 8 * 32 + 1, 1, 9, 0, 0, 0, False, 2), (300, 136, 8 * 32 + 6, 2, 9, 0, 0, 0, False, 0),
    ]
    if not small:
        cases += [(768, 256, 64 * 30 * 4, 6, 32, 0, 30, 256, False, 0), (768, 512, 48 * 30 * 2, 2, 16, 0, 30, 0, False, 1), (200, 264, 4095, 3, 24, 7, 0, 0, True, 0),
                  (768, 256, 64 * 30 * 4, 6, 32, 0, 30, 256, False, 1)]
    for (M, Nn, K, n, sk, sega, segb, gap, acc, opt) in cases:
        Mw = M + gap
        gap_at = (M // 2) // 4 * 4 if gap else 0

        def mk(width, seg):
            """(dense values (K, width), stored tensor, Operand fields): seg != 0 stores rows as (K / seg, seg + 2, width + 6) with the
            view starting at slot 1 -- the (B, T + 2, 2H) sequence layout of the engine."""
            v = (rng.standard_normal((K, width)) * np.exp(rng.uniform(-6, 6, (K, 1)))).astype(np.float32)
            if not seg:
                return v, v, dict(ld=width)
            assert K % seg == 0
            st = rng.standard_normal((K // seg, seg + 2, width + 6)).astype(np.float32)
            st[:, 1:seg + 1, 2:2 + width] = v.reshape(K // seg, seg, width)
            return v, st, dict(ld=width + 6, seg=seg, seg_stride=(seg + 2) * (width + 6), off=(width + 6) + 2)
        As, Bs = [mk(Mw, sega) for _ in range(n)], [mk(Nn, segb) for _ in range(n)]
        C0 = rng.standard_normal((n, M, Nn)).astype(np.float32)
        At, Bt, C = [T_(a[1], dev) for a in As], [T_(b[1], dev) for b in Bs], T_(C0, dev)
        ws = torch.zeros(n * sk * M * Nn, device=dev)
        opA = [Operand(t, **a[2]) for t, a in zip(At, As)]
        opB = [Operand(t, **b[2]) for t, b in zip(Bt, Bs)]
        assert ops.gemm_split_ok(M, Nn, K, opA, opB, sk, gap_at, gap)
        ops.gemm_group(M, Nn, K, opA, 1, opB, 1, C, [g * M * Nn for g in range(n)], Nn, sk, ws, accumulate=acc, a_gap_at=gap_at, a_gap=gap, split=opt)
        out = N_(C)
        cols = np.r_[0:gap_at, gap_at + gap:Mw] if gap else np.arange(M)
        for g in range(n):
            a, b = As[g][0][:, cols], Bs[g][0]
            base = C0[g].astype(np.float64) if acc else 0.0
            mag = np.abs(a).astype(np.float64).T @ np.abs(b).astype(np.float64)           # sum |a||b| per output
            six = split_reference(a, b) + base
            exact = a.astype(np.float64).T @ b.astype(np.float64) + base
            tag = str((M, Nn, K, n, sk, sega, segb, gap, acc, opt, g))
            # fp32 accumulation of the leading product alone (two accumulators: measured <= 2.2e-7 of sum |a||b| on the MI355X) or of all six
            # products (one: <= 3.6e-7, the f32-input MFMA kernel's own level); the dropped plane products add < 2^-21
            tol = 2.0 ** -21 if (opt & 3) != 1 else 2.0 ** -20
            assert np.all(np.abs(out[g] - six) <= tol * (mag + np.abs(base)) + 1e-30), tag + f" vs six planes: {np.max(np.abs(out[g] - six) / (mag + 1e-30)):.3e}"
            assert np.all(np.abs(out[g] - exact) <= 1.5 * tol * (mag + np.abs(base)) + 1e-30), tag + f" vs exact: {np.max(np.abs(out[g] - exact) / (mag + 1e-30)):.3e}"
    # every problem naming one C: the partial sums of all problems are one stack
    M, Nn, K, n, sk = 64, 66, 8 * 32, 3, 8
    As = [rng.standard_normal((K, M)).astype(np.float32) for _ in range(n)]
    Bs = [rng.standard_normal((K, Nn)).astype(np.float32) for _ in range(n)]
    C = torch.zeros(M, Nn, device=dev)
    ws = torch.zeros(n * sk * M * Nn, device=dev)
    ops.gemm_group(M, Nn, K, [Operand(T_(a, dev), M) for a in As], 1, [Operand(T_(b, dev), Nn) for b in Bs], 1, C, [0] * n, Nn, sk, ws, split=0)
    ref = sum(a.astype(np.float64).T @ b.astype(np.float64) for a, b in zip(As, Bs))
    np.testing.assert_allclose(N_(C), ref, atol=1e-5 * np.sqrt(K * n))
    # what it refuses: odd pitches / misaligned operands (8-byte loads), fewer than 8 slabs
    a, b = torch.zeros(K, 65, device=dev), torch.zeros(K, 64, device=dev)
    assert not ops.gemm_split_ok(64, 64, K, [Operand(a, 65)], [Operand(b, 64)], 8)
    assert not ops.gemm_split_ok(64, 64, K, [Operand(a, 64, off=1)], [Operand(b, 64)], 8)
    with pytest.raises(Exception):
        ops.gemm_group(64, 64, K, [Operand(a, 65)], 1, [Operand(b, 64)], 1, C, [0], 64, 8, ws, split=0)


def check_gemm_split_rows(dev, small=True):
    """vame_gemm_bf16x6_f32 (the split-bf16 contraction of a row-major A with a weight matrix, no split-K) against float64: ragged M / N
    tiles, two-level (batch, time) rows of A inside a wider padded sequence buffer, row-major and k-major B inside wider rows, bias,
    accumulation, a C with a pitch and an offset, both accumulator options.  Tolerances as check_gemm_split."""
    rng = np.random.default_rng(12)
    cases = [  # M, N, K, seg_a, b_kmajor, bias, acc, opt
        (70, 136, 64, 0, 0, True, False, 0), (128, 72, 96, 0, 1, False, True, 1), (6 * 30, 100, 64, 30, 0, True, False, 1),
        (9 * 15, 130, 160, 15, 1, False, False, 0), (301, 30, 32, 7, 0, False, True, 0), (257, 258, 128, 0, 1, True, True, 2), (33, 137, 64, 0, 0, True, False, 1),
    ]
    if not small:
        cases += [(64 * 30, 768, 512, 30, 0, True, False, 0), (64 * 30, 512, 768, 0, 1, False, True, 0), (4001, 770, 512, 0, 0, True, False, 1),
                  (48 * 30, 1024, 1536, 30, 1, False, False, 1), (40 * 30, 1536, 1024, 30, 0, True, False, 0)]
    for (M, Nn, K, sega, bkm, with_bias, acc, opt) in cases:
        a = (rng.standard_normal((M, K)) * np.exp(rng.uniform(-6, 6, (M, 1)))).astype(np.float32)
        if sega:        # rows (b, t) stored as (M / seg, seg + 2, K + 8) with the view at slot 1, column 4: the engine's (B, T + 2, 2H) layout
            st = rng.standard_normal((M // sega, sega + 2, K + 8)).astype(np.float32)
            st[:, 1:sega + 1, 4:4 + K] = a.reshape(M // sega, sega, K)
            opA = Operand(T_(st, dev), K + 8, off=(K + 8) + 4, seg=sega, seg_stride=(sega + 2) * (K + 8))
        else:
            st = rng.standard_normal((M, K + 4)).astype(np.float32)
            st[:, :K] = a
            opA = Operand(T_(st, dev), K + 4)
        if bkm:         # B[k][n] inside rows of N + 6
            b = rng.standard_normal((K, Nn)).astype(np.float32)
            sb = rng.standard_normal((K, Nn + 6)).astype(np.float32)
            sb[:, 2:2 + Nn] = b
            opB = Operand(T_(sb, dev), Nn + 6, off=2)
            bt = b
        else:           # B[n][k] inside rows of K + 4
            b = rng.standard_normal((Nn, K)).astype(np.float32)
            sb = rng.standard_normal((Nn, K + 4)).astype(np.float32)
            sb[:, 4:] = b
            opB = Operand(T_(sb, dev), K + 4, off=4)
            bt = b.T
        bias = rng.standard_normal(Nn).astype(np.float32) if with_bias else None
        ldc = Nn + 3
        C0 = rng.standard_normal((M + 1, ldc)).astype(np.float32)
        C = T_(C0, dev)
        assert ops.gemm_split_rows_ok(M, Nn, K, opA, 0, opB, bkm)
        ops.gemm(M, Nn, K, opA, 0, opB, bkm, C, ldc, c_off=ldc + 1, bias=T_(bias, dev) if with_bias else None, accumulate=acc, split=opt)
        full = N_(C)
        out = full[1:, 1:1 + Nn]
        keep = np.ones_like(C0, dtype=bool)
        keep[1:, 1:1 + Nn] = False
        assert np.array_equal(full[keep], C0[keep]), "wrote outside C"
        base = (C0[1:, 1:1 + Nn].astype(np.float64) if acc else 0.0) + (bias.astype(np.float64)[None, :] if with_bias else 0.0)
        mag = np.abs(a).astype(np.float64) @ np.abs(bt).astype(np.float64)
        six = split_reference(np.ascontiguousarray(a.T), np.ascontiguousarray(bt)) + base
        exact = a.astype(np.float64) @ bt.astype(np.float64) + base
        tag = str((M, Nn, K, sega, bkm, with_bias, acc, opt))
        tol = 2.0 ** -21 if (opt & 3) != 1 else 2.0 ** -20
        assert np.all(np.abs(out - six) <= tol * (mag + np.abs(base)) + 1e-30), tag + f" vs six planes: {np.max(np.abs(out - six) / (mag + 1e-30)):.3e}"
        assert np.all(np.abs(out - exact) <= 1.5 * tol * (mag + np.abs(base)) + 1e-30), tag + f" vs exact: {np.max(np.abs(out - exact) / (mag + 1e-30)):.3e}"
    # what it refuses: K not a multiple of 32, a k-major A, misaligned / odd-pitch operands
    a, b, C = torch.zeros(64, 72, device=dev), torch.zeros(64, 72, device=dev), torch.zeros(64, 64, device=dev)
    assert not ops.gemm_split_rows_ok(64, 64, 40, Operand(a, 72), 0, Operand(b, 72), 0)
    assert not ops.gemm_split_rows_ok(64, 64, 64, Operand(a, 72), 1, Operand(b, 72), 0)
    assert not ops.gemm_split_rows_ok(64, 64, 64, Operand(a, 72, off=2), 0, Operand(b, 72), 0)
    assert not ops.gemm_split_rows_ok(64, 63, 64, Operand(a, 72), 0, Operand(b, 71), 1)
    with pytest.raises(Exception):
        ops.gemm(64, 64, 40, Operand(a, 72), 0, Operand(b, 72), 0, C, 64, split=0)
    with pytest.raises(Exception):
        ops.gemm(64, 64, 64, Operand(a, 72, off=2), 0, Operand(b, 72), 0, C, 64, split=0)


def check_gemm_group_shared_output(dev):
    """All problems of a group naming one C: C (+)= sum_g A_g B_g (dz of the decoders), against float64."""
    rng = np.random.default_rng(4)
    for (M, Nn, K, n, acc) in [(100, 30, 256, 4, False), (64, 30, 512, 2, True), (130, 72, 768, 3, True)]:
        As = [rng.standard_normal((M, K)).astype(np.float32) for _ in range(n)]
        Bs = [rng.standard_normal((K, Nn)).astype(np.float32) for _ in range(n)]
        C0 = rng.standard_normal((M, Nn)).astype(np.float32)
        At, Bt, C = [T_(a, dev) for a in As], [T_(b, dev) for b in Bs], T_(C0, dev)
        ws = torch.zeros(n * 8 * M * Nn, device=dev)
        ops.gemm_group(M, Nn, K, [Operand(a, K) for a in At], 0, [Operand(b, Nn) for b in Bt], 1, C, [0] * n, Nn, 8, ws, accumulate=acc)
        ref = sum(a.astype(np.float64) @ b.astype(np.float64) for a, b in zip(As, Bs)) + (C0 if acc else 0)
        np.testing.assert_allclose(N_(C), ref, atol=2e-4 * np.sqrt(K * n))


def check_gemm_cases(dev, small=True):
    rng = np.random.default_rng(0)
    cases = [  # M, N, K, akm, bkm, splitk, bias, acc
        (70, 40, 50, 0, 0, 1, True, False), (30, 24, 30, 0, 0, 1, True, True), (45, 130, 37, 0, 1, 1, False, False),
        (33, 30, 200, 1, 1, 3, False, True), (96, 136, 64, 1, 1, 1, True, False), (130, 24, 19, 0, 0, 2, True, False),
    ]
    if not small:
        cases += [(512, 768, 512, 0, 0, 1, True, False), (768, 256, 4096, 1, 1, 8, False, False), (1000, 512, 768, 0, 1, 1, False, True)]
    cases = [c + (3,) for c in cases] + [c + (4,) for c in cases]      # row pad 3: scalar stores; 4: 16-byte aligned rows (vector stores)
    for (M, Nn, K, akm, bkm, sk, hb, acc, pad) in cases:
        A = rng.standard_normal((K, M) if akm else (M, K)).astype(np.float32)
        Bm = rng.standard_normal((K, Nn) if bkm else (Nn, K)).astype(np.float32)
        bias = rng.standard_normal(Nn).astype(np.float32) if hb else None
        C0 = rng.standard_normal((M, Nn + pad)).astype(np.float32)
        At, Bt, Ct = T_(A, dev), T_(Bm, dev), T_(C0, dev)
        ws = torch.zeros(sk * M * Nn, device=dev) if sk > 1 else None
        ops.gemm(M, Nn, K, Operand(At, A.shape[1]), akm, Operand(Bt, Bm.shape[1]), bkm, Ct, Nn + pad, bias=T_(bias, dev) if hb else None,
                 accumulate=acc, splitk=sk, ws=ws)
        ref = (A.T if akm else A).astype(np.float64) @ (Bm if bkm else Bm.T).astype(np.float64)
        if hb:
            ref += bias
        if acc:
            ref += C0[:, :Nn]
        out = N_(Ct)
        np.testing.assert_allclose(out[:, :Nn], ref, atol=2e-4 * max(1, np.sqrt(K)), err_msg=str((M, Nn, K, akm, bkm, sk)))
        np.testing.assert_array_equal(out[:, Nn:], C0[:, Nn:])
    # two-level row addressing: rows (b,t) of a padded (B,T+2,W) sequence, and a time-constant operand
    Bq, Tq, W, Nn = 6, 5, 20, 12
    Y = rng.standard_normal((Bq, Tq + 2, W)).astype(np.float32)
    Wt = rng.standard_normal((Nn, W)).astype(np.float32)
    C = torch.zeros(Bq * Tq, Nn, device=dev)
    Yt = T_(Y, dev)
    ops.gemm(Bq * Tq, Nn, W, Operand(Yt, W, off=W, seg=Tq, seg_stride=(Tq + 2) * W), 0, Operand(T_(Wt, dev), W), 0, C, Nn)
    np.testing.assert_allclose(N_(C), (Y[:, 1:Tq + 1].reshape(-1, W) @ Wt.T), atol=1e-4)
    dG = rng.standard_normal((Bq * Tq, 16)).astype(np.float32)
    z = rng.standard_normal((Bq, 10)).astype(np.float32)
    C = torch.zeros(16, 10, device=dev)
    ops.gemm(16, 10, Bq * Tq, Operand(T_(dG, dev), 16), 1, Operand(T_(z, dev), 0, seg=Tq, seg_stride=10), 1, C, 10)
    np.testing.assert_allclose(N_(C), dG.T @ np.repeat(z, Tq, 0), atol=1e-4)
    # column gap on the k-major A operand: rows of C come from columns [0,8) and [12,16) of dG
    C = torch.zeros(12, 10, device=dev)
    ops.gemm(12, 10, Bq * Tq, Operand(T_(dG, dev), 16), 1, Operand(T_(z, dev), 0, seg=Tq, seg_stride=10), 1, C, 10, a_gap_at=8, a_gap=4)
    np.testing.assert_allclose(N_(C), np.concatenate([dG[:, :8], dG[:, 12:]], 1).T @ np.repeat(z, Tq, 0), atol=1e-4)


def check_gemm_pipelined_shapes(dev):
    """Shapes that qualify for the software-pipelined loop of gemm_kernel (interior 128x128 tiles, >= 3 whole k-tiles, 16-byte
    aligned operands): all three layouts, split-K, two-level row addressing with one segment border per k-tile, the column gap."""
    rng = np.random.default_rng(5)
    for (M, Nn, K, akm, bkm, sk) in [(256, 128, 160, 0, 0, 1), (128, 256, 96, 0, 1, 1), (256, 256, 128, 1, 1, 1), (128, 128, 1024, 1, 1, 8),
                                     (384, 128, 100, 0, 0, 1)]:
        A = rng.standard_normal((K, M) if akm else (M, K)).astype(np.float32)
        Bm = rng.standard_normal((K, Nn) if bkm else (Nn, K)).astype(np.float32)
        Ct = torch.zeros(M, Nn, device=dev)
        ws = torch.zeros(sk * M * Nn, device=dev) if sk > 1 else None
        ops.gemm(M, Nn, K, Operand(T_(A, dev), A.shape[1]), akm, Operand(T_(Bm, dev), Bm.shape[1]), bkm, Ct, Nn, splitk=sk, ws=ws)
        ref = (A.T if akm else A).astype(np.float64) @ (Bm if bkm else Bm.T).astype(np.float64)
        np.testing.assert_allclose(N_(Ct), ref, atol=2e-4 * np.sqrt(K), err_msg=str((M, Nn, K, akm, bkm, sk)))
    # weight-gradient form on padded sequences: A = dG (B*T, 4H) with a column gap, B = rows (b,t) of a (B, T+2, W) sequence
    Bq, Tq, Hh = 4, 40, 64                                 # K = B*T = 160 = 5 k-tiles, segment length 40 >= 32
    dG = rng.standard_normal((Bq * Tq, 4 * Hh)).astype(np.float32)
    Y = rng.standard_normal((Bq, Tq + 2, 128)).astype(np.float32)
    C = torch.zeros(3 * Hh + 64, 128, device=dev)         # 256 x 128 output: rows from columns [0,128) and [192,320) of dG
    ops.gemm(256, 128, Bq * Tq, Operand(T_(dG[:, :1].repeat(1, 1) if False else np.concatenate([dG, dG[:, :64]], 1), dev), 320), 1,
             Operand(T_(Y, dev), 128, off=128, seg=Tq, seg_stride=(Tq + 2) * 128), 1, C, 128, a_gap_at=128, a_gap=64)
    dGw = np.concatenate([dG, dG[:, :64]], 1)
    Aeff = np.concatenate([dGw[:, :128], dGw[:, 192:320]], 1)
    np.testing.assert_allclose(N_(C), Aeff.T.astype(np.float64) @ Y[:, 1:Tq + 1].reshape(-1, 128).astype(np.float64), atol=3e-3)
    # NT form with a segmented row-major A (the gi projection reads rows (b,t) of a padded sequence)
    Bq, Tq, W = 8, 16, 96                                  # M = 128 rows, K = 96
    Y = rng.standard_normal((Bq, Tq + 2, W)).astype(np.float32)
    Wt = rng.standard_normal((128, W)).astype(np.float32)
    C = torch.zeros(Bq * Tq, 128, device=dev)
    ops.gemm(Bq * Tq, 128, W, Operand(T_(Y, dev), W, off=W, seg=Tq, seg_stride=(Tq + 2) * W), 0, Operand(T_(Wt, dev), W), 0, C, 128)
    np.testing.assert_allclose(N_(C), Y[:, 1:Tq + 1].reshape(-1, W).astype(np.float64) @ Wt.T.astype(np.float64), atol=1e-3)


def _gru_weights(rng, I, H):
    k = 1 / np.sqrt(H)
    return (rng.uniform(-k, k, (3 * H, I)).astype(np.float32), rng.uniform(-k, k, (3 * H, H)).astype(np.float32),
            rng.uniform(-k, k, 3 * H).astype(np.float32), rng.uniform(-k, k, 3 * H).astype(np.float32))


def _pack(dev, W_hh, b_ih, b_hh, H):
    wpf = torch.zeros(3 * H * H, device=dev)
    wpb = torch.zeros(3 * H * H, device=dev)
    bgi = torch.zeros(3 * H, device=dev)
    bhn = torch.zeros(H, device=dev)
    ops.gru_pack(T_(W_hh, dev), T_(b_ih, dev), T_(b_hh, dev), H, wpf, wpb, bgi, bhn)
    return wpf, wpb, bgi, bhn


FORCE_WIDE = False           # set by check_gru_wide_small: the two-blocks-per-wave kernels at H <= 256


def run_gru_fwd(dev, H, B, T, seed=0, coop=None, coop_chunks=None, want_rows=False, coop_kernel=None):
    """Two streams (forward + reverse dir, with h0) in one launch; returns everything needed for bwd.
    coop: an ops.CoopState -> the column-split small-batch kernel instead of the batch-tile-persistent one."""
    rng = np.random.default_rng(seed)
    I = 7
    x = rng.standard_normal((B, T, I)).astype(np.float32)
    st = []
    Y = torch.zeros(B, T + 2, 2 * H, device=dev)
    hN = torch.zeros(B, 2 * H, device=dev)
    rows = []
    for d in range(2):
        W_ih, W_hh, b_ih, b_hh = _gru_weights(rng, I, H)
        h0 = rng.standard_normal((B, H)).astype(np.float32) * 0.5 if d == 1 else None
        wpf, wpb, bgi, bhn = _pack(dev, W_hh, b_ih, b_hh, H)
        gi = T_((x.reshape(B * T, I) @ W_ih.T + N_(bgi)).reshape(B, T, 3 * H), dev)
        # NaN-filled: every stash entry of a tile (also of its rows past the batch, which BPTT multiplies by zero) must be written
        stash = torch.full((ops.gru_stash_floats(B, T, H),), float("nan"), device=dev)
        h0t = T_(h0, dev) if h0 is not None else None
        rows.append({GF["GI"]: ops.addr(gi), GF["GI_ROW"]: T * 3 * H, GF["GI_T"]: 3 * H, GF["WP"]: ops.addr(wpf),
                     GF["BHN"]: ops.addr(bhn), GF["H0"]: ops.addr(h0t), GF["H0_ROW"]: H,
                     GF["Y"]: ops.addr(Y, 2 * H + d * H), GF["Y_ROW"]: (T + 2) * 2 * H, GF["Y_T"]: 2 * H,
                     GF["HN"]: ops.addr(hN, d * H), GF["HN_ROW"]: 2 * H, GF["STASH"]: ops.addr(stash), GF["T"]: T,
                     GF["REVERSE"]: d, GF["PAD"]: 1})
        st.append(dict(W_ih=W_ih, W_hh=W_hh, b_ih=b_ih, b_hh=b_hh, h0=h0, wpb=wpb, stash=stash, keep=(gi, wpf, bhn, h0t)))
    launch_gru_fwd(rows, B, H, coop, coop_chunks, coop_kernel)
    if want_rows:
        return x, st, Y, hN, rows
    return x, st, Y, hN


FWD_KERNEL = ops.KERNEL_AUTO      # kernel argument of launch_gru_fwd's gru_seq_fwd launches (GF_OPT descriptor field)


def launch_gru_fwd(rows, B, H, coop=None, coop_chunks=None, coop_kernel=None):
    if coop is not None:
        for chunk in (coop_chunks or [(0, 0)]):
            ops.gru_coop_fwd(rows, B, H, coop, rows=chunk, kernel=coop_kernel or ops.KERNEL_AUTO)
    elif H > 256 or FORCE_WIDE:
        ops.gru_wide_fwd(rows, B, H, kernel=FWD_KERNEL)
    else:
        ops.gru_seq_fwd(rows, B, H, kernel=FWD_KERNEL)


def check_gru_fwd_ring_stress(dev, H, B, T, launches=200):
    """Repeat / stress guard of the forward kernels' weight ring (inline-asm loads with hand-counted vmcnt waits: exactly what the host
    emulator cannot see, and a memory-ordering slip there shows only under load): the SAME launch `launches` times while a side
    stream streams HBM at full rate, outputs compared bit for bit -- h sequence + final state after EVERY launch, the BPTT stash
    every tenth -- against the first launch, which is itself checked against the oracle by check_gru_fwd at a small batch."""
    x, st, Y, hN, rows = run_gru_fwd(dev, H, B, T, want_rows=True)
    torch.cuda.synchronize()
    Y0, hN0, st0 = Y.clone(), hN.clone(), [s["stash"].clone() for s in st]
    side = torch.cuda.Stream()
    big = [torch.empty(1 << 28, device=dev) for _ in range(2)]             # 2 x 1 GiB, copied back and forth beside the launches
    stop = torch.cuda.Event()
    with torch.cuda.stream(side):
        for i in range(launches):
            big[(i + 1) & 1].copy_(big[i & 1])
        stop.record()
    bad = torch.zeros(3, dtype=torch.int64, device=dev)
    for it in range(launches):
        Y.zero_()
        launch_gru_fwd(rows, B, H)
        bad[0] += (Y != Y0).any()
        bad[1] += (hN != hN0).any()
        if it % 10 == 9:
            for s, s0 in zip(st, st0):
                bad[2] += (s["stash"] != s0).any()
    overlapped = not stop.query()                 # the background copies were still running when the last launch was enqueued
    torch.cuda.synchronize()
    assert bad.tolist() == [0, 0, 0], f"launch-to-launch differences (Y, hN, stash): {bad.tolist()}"
    return overlapped


def check_gru_coop_fwd(dev, H, B, T, launches=3):
    """Column-split kernel vs the batch-tile-persistent one: outputs, final state and stash agree to summation-order rounding (the
    column-split kernel sums K = H in two halves of 16 x 16 x 4 MFMAs), it agrees with the oracle, and ITS results are the same bits
    whichever form a launch takes -- whole batch in 16-row groups, 32-row groups (kernel option), two row-range launches -- over
    repeated launches that reuse the flag words (epoch logic)."""
    assert ops.gru_coop_supported(2, B, H) and not ops.gru_coop_supported(2, 8192, H)
    state = ops.CoopState(torch.device(dev))
    state.epoch[0] = -T - 3                               # (= 2^32 - T - 3 as unsigned) the second launch crosses the 2^32 wrap of the flag epoch
    state.flags.fill_(-T - 4)                             # ... as left behind by a launch just before it
    x, st0, Y0, hN0 = run_gru_fwd(dev, H, B, T)
    forms = [(None, ops.KERNEL_AUTO), (None, ops.KERNEL_LOCKSTEP)]
    if B > 32:
        forms.append(([(0, 32), (32, B - 32)], ops.KERNEL_AUTO))         # whole batch, then two row-range launches
    ntiles, NW = (B + 31) // 32, H // 32
    q, lane, e = np.meshgrid(np.arange(4), np.arange(64), np.arange(4), indexing="ij")
    r = 4 * q + e
    row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)                                   # CR(r) + 4 * (lane >> 5)
    valid = (np.arange(ntiles)[:, None, None, None] * 32 + row[None]) < B              # (ntiles, 4, 64, 4)
    valid = np.broadcast_to(valid[:, None, None, None], (ntiles, T, NW, 5, 4, 64, 4)).reshape(-1)
    first = None
    for it in range(launches):
        chunks, kern = forms[it % len(forms)]
        x, st, Y, hN = run_gru_fwd(dev, H, B, T, coop=state, coop_chunks=chunks, coop_kernel=kern)
        np.testing.assert_allclose(N_(Y), N_(Y0), rtol=0, atol=2e-6)
        np.testing.assert_allclose(N_(hN), N_(hN0), rtol=0, atol=2e-6)
        # stash entries of rows past the batch (last tile) are don't-cares of both kernels: compare the valid rows
        for a, b in zip(st, st0):
            sa, sb = N_(a["stash"]), N_(b["stash"])
            np.testing.assert_allclose(sa[valid], sb[valid], rtol=0, atol=4e-6)      # (the fifth quantity is a pre-activation of a few units)
            assert np.isfinite(sa).all()
        got = [N_(Y), N_(hN)] + [N_(a["stash"])[valid] for a in st]
        if first is None:
            first = got
        for g_, f_ in zip(got, first):
            np.testing.assert_array_equal(g_, f_)
    assert int(state.status.item()) == 0
    check_gru_fwd(dev, H, B, T)


def check_gru_fwd(dev, H, B, T):
    x, st, Y, hN = run_gru_fwd(dev, H, B, T)
    Yn, hNn = N_(Y), N_(hN)
    for d, s in enumerate(st):
        out, hn, _ = vo.gru_dir_forward(x, s["h0"], s["W_ih"], s["W_hh"], s["b_ih"], s["b_hh"], reverse=bool(d))
        np.testing.assert_allclose(Yn[:, 1:T + 1, d * H:(d + 1) * H], out, atol=2e-5)
        np.testing.assert_allclose(hNn[:, d * H:(d + 1) * H], hn, atol=2e-5)
        pad = Yn[:, T + 1 if d else 0, d * H:(d + 1) * H]
        np.testing.assert_allclose(pad, s["h0"] if s["h0"] is not None else 0 * pad, atol=0)


def check_gru_wide(dev, H, B, T):
    """256 < H <= 512: the two-blocks-per-wave persistent forward and three-phase BPTT kernels (gru_wide.hip) vs the oracle, and one BPTT
    step of the step-wise backward gate kernel reading the fragment-order stash vs the coefficients the oracle's forward cache implies."""
    check_gru_fwd(dev, H, B, T)
    check_gru_bwd(dev, H, B, T)
    x, st, Y, hN = run_gru_fwd(dev, H, B, T)
    rng = np.random.default_rng(3)
    for d, s_ in enumerate(st):
        out, hn, cache = vo.gru_dir_forward(x, s_["h0"], s_["W_ih"], s_["W_hh"], s_["b_ih"], s_["b_hh"], reverse=bool(d))
        for (t, r, u, n, ghn, hprev) in (cache[0], cache[-1]):
            dh0 = rng.standard_normal((B, H)).astype(np.float32)
            dy = rng.standard_normal((B, H)).astype(np.float32)
            dh, dG, dgh = T_(dh0, dev), torch.zeros(B, 4 * H, device=dev), torch.zeros(B, 3 * H, device=dev)
            ops.gru_cell_bwd_frag(s_["stash"], T, t, dh, T_(dy, dev), 0, H, dG, 0, 4 * H, dgh, B, H)
            dd = dh0 + dy
            dan = dd * (1 - u) * (1 - n * n)
            dau = dd * (hprev - n) * u * (1 - u)
            dghn = dan * r
            dar = dghn * ghn * (1 - r)
            np.testing.assert_allclose(N_(dG), np.concatenate([dar, dau, dan, dghn], 1), atol=3e-5)
            np.testing.assert_allclose(N_(dgh), np.concatenate([dar, dau, dghn], 1), atol=3e-5)
            np.testing.assert_allclose(N_(dh), dd * u, atol=3e-5)


def check_gru_wide_small(dev, H, B, T):
    """The wide kernels are also instantiated for H = 128, 192, 256 (same contract as the gru_seq kernels there)."""
    global FORCE_WIDE
    FORCE_WIDE = True
    try:
        check_gru_fwd(dev, H, B, T)
        check_gru_bwd(dev, H, B, T)
    finally:
        FORCE_WIDE = False


def check_gru_fwd_fused(dev, H, B, T, I=24):
    """Fused input projection (encoder layer 0 mode): x (B,L,F) windows are read directly, no gi tensor."""
    rng = np.random.default_rng(11)
    L = T + 3
    win = rng.standard_normal((B, L, I)).astype(np.float32)
    wint = T_(win, dev)
    Y = torch.zeros(B, T + 2, 2 * H, device=dev)
    hN = torch.zeros(B, 2 * H, device=dev)
    rows, st, keep = [], [], []
    for d in range(2):
        W_ih, W_hh, b_ih, b_hh = _gru_weights(rng, I, H)
        wpf, wpb, bgi, bhn = _pack(dev, W_hh, b_ih, b_hh, H)
        wpx = torch.zeros(3 * H * 32, device=dev)
        ops.gru_pack_x(T_(W_ih, dev), I, H, wpx)
        stash = torch.zeros(ops.gru_stash_floats(B, T, H), device=dev)
        rows.append({GF["GI"]: ops.addr(wint), GF["GI_ROW"]: L * I, GF["GI_T"]: I, GF["WP"]: ops.addr(wpf), GF["BHN"]: ops.addr(bhn),
                     GF["Y"]: ops.addr(Y, 2 * H + d * H), GF["Y_ROW"]: (T + 2) * 2 * H, GF["Y_T"]: 2 * H, GF["HN"]: ops.addr(hN, d * H),
                     GF["HN_ROW"]: 2 * H, GF["STASH"]: ops.addr(stash), GF["T"]: T, GF["REVERSE"]: d, GF["PAD"]: 1,
                     GF["WPX"]: ops.addr(wpx), GF["BGI"]: ops.addr(bgi), GF["XF"]: I})
        st.append((W_ih, W_hh, b_ih, b_hh))
        keep.append((wpf, wpb, bgi, bhn, wpx, stash))
    ops.gru_seq_fwd(rows, B, H, kernel=FWD_KERNEL)
    Yn, hNn = N_(Y), N_(hN)
    for d, (W_ih, W_hh, b_ih, b_hh) in enumerate(st):
        out, hn, _ = vo.gru_dir_forward(win[:, :T], None, W_ih, W_hh, b_ih, b_hh, reverse=bool(d))
        np.testing.assert_allclose(Yn[:, 1:T + 1, d * H:(d + 1) * H], out, atol=2e-5)
        np.testing.assert_allclose(hNn[:, d * H:(d + 1) * H], hn, atol=2e-5)
    return Y, hN, [k[5] for k in keep]


def _valid_stash_mask(B, T, H):
    """Stash entries of rows past the batch (last tile) are don't-cares: True for the entries of valid rows."""
    ntiles, NW = (B + 31) // 32, H // 32
    q, lane, e = np.meshgrid(np.arange(4), np.arange(64), np.arange(4), indexing="ij")
    r = 4 * q + e
    row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)                                   # CR(r) + 4 * (lane >> 5)
    valid = (np.arange(ntiles)[:, None, None, None] * 32 + row[None]) < B              # (ntiles, 4, 64, 4)
    return np.broadcast_to(valid[:, None, None, None], (ntiles, T, NW, 5, 4, 64, 4)).reshape(-1)


def check_gru_wide_skew_fwd(dev, H, B, T, force_wide=False):
    """The skewed form of the two-blocks-per-wave forward kernel (gru_wide.hip: gru_wide_skew_fwd_kernel) against the oracle and bit for bit
    against the lock-step wide kernel: per-step gi streams with and without an initial state, both directions."""
    global FWD_KERNEL, FORCE_WIDE
    outs = {}
    try:
        FORCE_WIDE = force_wide
        for kern in (ops.KERNEL_SKEWED, ops.KERNEL_LOCKSTEP):
            FWD_KERNEL = kern
            if kern == ops.KERNEL_SKEWED:
                check_gru_fwd(dev, H, B, T)
            x, st, Y, hN = run_gru_fwd(dev, H, B, T, seed=2)
            outs[kern] = [Y, hN] + [s_["stash"] for s_ in st]
    finally:
        FWD_KERNEL, FORCE_WIDE = ops.KERNEL_AUTO, False
    valid = torch.from_numpy(_valid_stash_mask(B, T, H).copy()).to(dev)
    for i, (a, b) in enumerate(zip(outs[ops.KERNEL_SKEWED], outs[ops.KERNEL_LOCKSTEP])):
        if a.dim() == 1:
            assert torch.equal(a[valid], b[valid]), f"stash {i} differs"
        else:
            assert torch.equal(a, b), f"output {i} differs: {float((a - b).abs().max())}"


def check_gru_skew_fwd(dev, H, B, T):
    """The skewed forward kernel (gru_seq.hip: gru_skew_fwd_kernel -- the two waves of a SIMD half a step apart) against the numpy oracle
    and BIT FOR BIT against the lock-step kernel (same arithmetic in the same order): h sequence, final state, BPTT stash -- for per-step gi
    streams with and without an initial state (both directions), the fused input projection, and a time-constant gi with h0 (decoder
    form).  The kernel is picked per launch by the GF_OPT descriptor field."""
    global FWD_KERNEL
    assert ops.gru_seq_fwd_has_kernel(H, ops.KERNEL_SKEWED)
    outs = {}
    try:
        for kern in (ops.KERNEL_SKEWED, ops.KERNEL_LOCKSTEP):
            FWD_KERNEL = kern
            if kern == ops.KERNEL_SKEWED:
                check_gru_fwd(dev, H, B, T)                       # vs the oracle
            x, st, Y, hN = run_gru_fwd(dev, H, B, T, seed=2)
            Yf, hNf, stf = check_gru_fwd_fused(dev, H, B, T)     # vs the oracle, fused input (zero initial state: skipped first K loop)
            # decoder form: gi constant in time, initial state given, one stream per direction, inference (no stash) on the reverse one
            rng = np.random.default_rng(4)
            W_ih, W_hh, b_ih, b_hh = _gru_weights(rng, 5, H)
            wpf, wpb, bgi, bhn = _pack(dev, W_hh, b_ih, b_hh, H)
            gi = T_(rng.standard_normal((B, 3 * H)).astype(np.float32), dev)
            h0 = T_(rng.standard_normal((B, H)).astype(np.float32) * 0.5, dev)
            Yc, hNc = torch.zeros(B, T + 2, 2 * H, device=dev), torch.zeros(B, 2 * H, device=dev)
            stc = torch.zeros(ops.gru_stash_floats(B, T, H), device=dev)
            rows = [{GF["GI"]: ops.addr(gi), GF["GI_ROW"]: 3 * H, GF["GI_T"]: 0, GF["WP"]: ops.addr(wpf), GF["BHN"]: ops.addr(bhn),
                     GF["H0"]: ops.addr(h0) if d == 0 else 0, GF["H0_ROW"]: H, GF["Y"]: ops.addr(Yc, 2 * H + d * H), GF["Y_ROW"]: (T + 2) * 2 * H,
                     GF["Y_T"]: 2 * H, GF["HN"]: ops.addr(hNc, d * H), GF["HN_ROW"]: 2 * H, GF["STASH"]: ops.addr(stc) if d == 0 else 0,
                     GF["T"]: T, GF["REVERSE"]: d, GF["PAD"]: 1} for d in range(2)]
            launch_gru_fwd(rows, B, H)
            outs[kern] = [Y, hN, Yf, hNf, Yc, hNc] + [s_["stash"] for s_ in st] + stf + [stc]
    finally:
        FWD_KERNEL = ops.KERNEL_AUTO
    valid = torch.from_numpy(_valid_stash_mask(B, T, H).copy()).to(dev)
    for i, (a, b) in enumerate(zip(outs[ops.KERNEL_SKEWED], outs[ops.KERNEL_LOCKSTEP])):
        if a.dim() == 1:                                          # a stash: compare the entries of rows inside the batch
            assert torch.equal(a[valid], b[valid]), f"stash {i} differs"
        else:
            assert torch.equal(a, b), f"output {i} differs: {float((a - b).abs().max())}"


def _run_gru_bwd(dev, H, B, T, st, Y, dYt, dhNt, coop=None, coop_chunks=None, coop_kernel=None):
    ntiles = (B + 31) // 32
    rows, outs = [], []
    for d, s in enumerate(st):
        dG = torch.zeros(B, T, 4 * H, device=dev)
        dh0 = torch.zeros(B, H, device=dev)
        dbias = torch.zeros(ntiles, 4 * H, device=dev)
        dgsum = torch.zeros(B, 3 * H, device=dev)
        rows.append({GB["STASH"]: ops.addr(s["stash"]), GB["Y"]: ops.addr(Y, 2 * H + d * H), GB["Y_ROW"]: (T + 2) * 2 * H,
                     GB["Y_T"]: 2 * H, GB["WPT"]: ops.addr(s["wpb"]), GB["DY"]: ops.addr(dYt, d * H), GB["DY_ROW"]: T * 2 * H,
                     GB["DY_T"]: 2 * H, GB["DHN"]: ops.addr(dhNt, d * H), GB["DHN_ROW"]: 2 * H, GB["DG"]: ops.addr(dG),
                     GB["DH0"]: ops.addr(dh0), GB["DH0_ROW"]: H, GB["DBIAS"]: ops.addr(dbias),
                     GB["T"]: T, GB["REVERSE"]: d, GB["PAD"]: 1})
        outs.append((dG, dh0, dbias, dgsum))
    if coop is not None:
        for chunk in (coop_chunks or [(0, 0)]):
            ops.gru_coop_bwd(rows, B, H, coop, rows=chunk, kernel=coop_kernel or ops.KERNEL_AUTO)
    elif H > 256 or FORCE_WIDE:
        ops.gru_wide_bwd(rows, B, H)
    else:
        ops.gru_seq_bwd(rows, B, H, kernel=BWD_KERNEL)
    return outs


def check_gru_coop_bwd(dev, H, B, T, launches=3):
    """Column-split BPTT kernel vs the batch-tile-persistent one on the same stash: dG, dh0 and the bias partial sums agree to
    summation-order rounding (the K = 3H contraction is split by member), over repeated launches sharing the flag words; the
    column-split results are the same bits in every form of the launch (16-row groups, 32-row groups, two row-range launches)."""
    x, st, Y, hN = run_gru_fwd(dev, H, B, T, seed=1)
    rng = np.random.default_rng(5)
    dYt = T_(rng.standard_normal((B, T, 2 * H)).astype(np.float32), dev)
    dhNt = T_(rng.standard_normal((B, 2 * H)).astype(np.float32), dev)
    ref = _run_gru_bwd(dev, H, B, T, st, Y, dYt, dhNt)
    state = ops.CoopState(torch.device(dev))
    forms = [(None, ops.KERNEL_AUTO), (None, ops.KERNEL_LOCKSTEP)]
    if B > 32:
        forms.append(([(0, 32), (32, B - 32)], ops.KERNEL_AUTO))
    first = None
    for it in range(launches):
        chunks, kern = forms[it % len(forms)]
        got = _run_gru_bwd(dev, H, B, T, st, Y, dYt, dhNt, coop=state, coop_chunks=chunks, coop_kernel=kern)
        for (dG, dh0, dbias, _), (dGr, dh0r, dbiasr, _) in zip(got, ref):
            np.testing.assert_allclose(N_(dG), N_(dGr), atol=2e-5 * float(np.abs(N_(dGr)).max()))
            np.testing.assert_allclose(N_(dh0), N_(dh0r), atol=2e-5 * float(np.abs(N_(dh0r)).max()))
            np.testing.assert_allclose(N_(dbias).sum(0), N_(dbiasr).sum(0), atol=1e-4 * float(np.abs(N_(dbiasr).sum(0)).max()))
        flat = [N_(t_) for o in got for t_ in o[:3]]
        if first is None:
            first = flat
        for a, b in zip(flat, first):
            np.testing.assert_array_equal(a, b)
    assert int(state.status.item()) == 0


BWD_KERNEL = ops.KERNEL_AUTO      # kernel argument of _run_gru_bwd's gru_seq_bwd launches (ops.KERNEL_*: a descriptor field, not process state)


def check_gru_ws_bwd(dev, H, B, T):
    """The wave-specialised BPTT kernel (gru_seq.hip: gru_ws_bwd_kernel; MFMA waves / memory waves) against the numpy oracle AND
    bit for bit against the lock-step kernel it replaces at H = 256 (same stash, same arithmetic order): dG, dh0; bias partials to rounding.
    The kernel is picked per launch by the GB_OPT descriptor field (ops.gru_seq_bwd(kernel=...))."""
    global BWD_KERNEL
    assert ops.gru_seq_bwd_has_kernel(H, ops.KERNEL_WS)
    try:
        BWD_KERNEL = ops.KERNEL_WS
        check_gru_bwd(dev, H, B, T)
        x, st, Y, hN = run_gru_fwd(dev, H, B, T, seed=1)
        rng = np.random.default_rng(5)
        dYt = T_(rng.standard_normal((B, T, 2 * H)).astype(np.float32), dev)
        dhNt = T_(rng.standard_normal((B, 2 * H)).astype(np.float32), dev)
        ws = _run_gru_bwd(dev, H, B, T, st, Y, dYt, dhNt)
        BWD_KERNEL = ops.KERNEL_LOCKSTEP
        ls = _run_gru_bwd(dev, H, B, T, st, Y, dYt, dhNt)
        if dev != "cpu":                        # hand-off races between the two roles would show as launch-to-launch differences
            BWD_KERNEL = ops.KERNEL_WS
            for _ in range(10):
                again = _run_gru_bwd(dev, H, B, T, st, Y, dYt, dhNt)
                assert all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) for a, b in zip(again, ws))
        for (dG, dh0, dbias, _), (dG2, dh02, dbias2, _) in zip(ws, ls):
            assert torch.equal(dG, dG2) and torch.equal(dh0, dh02)
            # bias partials: the same 16 x T terms per lane, summed pairwise (packed fp32 adds) instead of one by one
            np.testing.assert_allclose(N_(dbias), N_(dbias2), rtol=2e-5, atol=2e-5 * float(np.abs(N_(dbias2)).max()))
    finally:
        BWD_KERNEL = ops.KERNEL_AUTO


def check_gru_kernel_option_is_an_argument(dev):
    """The kernel choice is a field of the descriptor (GB_OPT), validated per call: an unknown value is refused, the wave-specialised
    kernel is refused for a hidden size it is not instantiated for, and the library reads no environment variable for it."""
    from vame_amd import _lib
    H, B, T = 96, 32, 2
    x, st, Y, hN = run_gru_fwd(dev, H, B, T, seed=1)
    dYt, dhNt = torch.zeros(B, T, 2 * H, device=dev), torch.zeros(B, 2 * H, device=dev)
    global BWD_KERNEL
    try:
        assert not ops.gru_seq_bwd_has_kernel(H, ops.KERNEL_WS) and ops.gru_seq_bwd_has_kernel(H, ops.KERNEL_LOCKSTEP)
        assert ops.gru_seq_bwd_has_kernel(192, ops.KERNEL_WS) and ops.gru_seq_fwd_has_kernel(192, ops.KERNEL_SKEWED) and not ops.gru_seq_fwd_has_kernel(96, ops.KERNEL_SKEWED)
        assert ops.gru_seq_bwd_has_kernel(256, ops.KERNEL_WS) and not ops.gru_seq_bwd_has_kernel(512, ops.KERNEL_AUTO)
        BWD_KERNEL = ops.KERNEL_WS
        with pytest.raises(_lib.VameHipError, match="not instantiated"):
            _run_gru_bwd(dev, H, B, T, st, Y, dYt, dhNt)
        BWD_KERNEL = 7
        with pytest.raises(_lib.VameHipError, match="unknown kernel option"):
            _run_gru_bwd(dev, H, B, T, st, Y, dYt, dhNt)
        # the cooperative launches know AUTO and LOCKSTEP (= 32-row groups) only
        state = ops.CoopState(torch.device(dev))
        x2, st2, Y2, hN2 = run_gru_fwd(dev, 128, 8, 2, seed=1)
        with pytest.raises(_lib.VameHipError, match="kernel option"):
            run_gru_fwd(dev, 128, 8, 2, seed=1, coop=state, coop_kernel=ops.KERNEL_SKEWED)
        with pytest.raises(_lib.VameHipError, match="kernel option"):
            _run_gru_bwd(dev, 128, 8, 2, st2, Y2, torch.zeros(8, 2, 256, device=dev), torch.zeros(8, 256, device=dev), coop=state, coop_kernel=ops.KERNEL_WS)
        assert int(state.status.item()) == 0
    finally:
        BWD_KERNEL = ops.KERNEL_AUTO


def check_gru_bwd(dev, H, B, T):
    x, st, Y, hN = run_gru_fwd(dev, H, B, T, seed=1)
    rng = np.random.default_rng(5)
    dY = rng.standard_normal((B, T, 2 * H)).astype(np.float32)
    dhN = rng.standard_normal((B, 2 * H)).astype(np.float32)
    dYt, dhNt = T_(dY, dev), T_(dhN, dev)
    outs = _run_gru_bwd(dev, H, B, T, st, Y, dYt, dhNt)
    for dG, _, _, dgsum in outs:
        ops.timesum(dG, B, T, 3 * H, 4 * H, dgsum)
    for d, s in enumerate(st):
        _, _, cache = vo.gru_dir_forward(x, s["h0"], s["W_ih"], s["W_hh"], s["b_ih"], s["b_hh"], reverse=bool(d))
        dx, dh0, dWi, dWh, dbi, dbh = vo.gru_dir_backward(x, cache, np.ascontiguousarray(dY[:, :, d * H:(d + 1) * H]),
                                                          dhN[:, d * H:(d + 1) * H], s["W_ih"], s["W_hh"])
        dG, dh0k, dbias, dgsum = [N_(o) for o in outs[d]]
        tol = 5e-5 * max(1.0, np.abs(dWh).max())
        np.testing.assert_allclose(dh0k, dh0, atol=5e-5)
        dgi = dG[:, :, :3 * H].reshape(B * T, 3 * H)
        np.testing.assert_allclose(dgi.T @ x.reshape(B * T, -1), dWi, atol=tol)
        np.testing.assert_allclose(dgi @ s["W_ih"], dx.reshape(B * T, -1), atol=5e-5)
        dbs = dbias.sum(0)
        tol_b = max(tol, 3e-5 * np.abs(dbi).max())            # fp32 sums over B*T terms: tolerance relative to the largest bias gradient
        np.testing.assert_allclose(dbs[:3 * H], dbi, atol=tol_b)
        np.testing.assert_allclose(np.concatenate([dbs[:2 * H], dbs[3 * H:]]), dbh, atol=tol_b)
        np.testing.assert_allclose(dgsum, dG[:, :, :3 * H].sum(1), atol=1e-5)
        # dW_hh from dG and the padded h sequence (h_prev = previous slot in step order)
        Yn = N_(Y)[:, :, d * H:(d + 1) * H]
        hprev = Yn[:, 2:T + 2] if d else Yn[:, 0:T]
        dgh = np.concatenate([dG[:, :, :2 * H], dG[:, :, 3 * H:]], 2).reshape(B * T, 3 * H)
        np.testing.assert_allclose(dgh.T @ hprev.reshape(B * T, H), dWh, atol=tol)


def check_gather(dev):
    rng = np.random.default_rng(2)
    F, Nn, L, B = 24, 300, 45, 9
    X = rng.standard_normal((F, Nn)).astype(np.float32)
    starts = rng.integers(0, Nn - L, B)
    out = torch.zeros(B, L, F, device=dev)
    ops.window_gather(T_(X, dev), Nn, F, T_(starts.astype(np.int64), dev), 0, B, L, out)
    np.testing.assert_array_equal(N_(out), vo.window_gather(X, starts, L))
    out2 = torch.zeros(7, 30, F, device=dev)
    ops.window_gather(T_(X, dev), Nn, F, None, 100, 7, 30, out2)
    np.testing.assert_array_equal(N_(out2), vo.window_gather(X, np.arange(100, 107), 30))


def check_latent(dev):
    rng = np.random.default_rng(3)
    B, Z = 37, 30
    mu, lvr, eps, dz = [rng.standard_normal((B, Z)).astype(np.float32) for _ in range(4)]
    for sp in (0, 1):
        for training in (1, 0):
            lv, z, kl = torch.zeros(B, Z, device=dev), torch.zeros(B, Z, device=dev), torch.zeros(1, device=dev)
            ops.latent_fwd(T_(mu, dev), T_(lvr, dev), T_(eps, dev), B, Z, sp, training, lv, z, kl)
            lv_ref = vo.softplus(lvr) if sp else lvr
            np.testing.assert_allclose(N_(lv), lv_ref, atol=1e-6)
            z_ref = eps * np.exp(0.5 * lv_ref) + mu if training else mu
            np.testing.assert_allclose(N_(z), z_ref, atol=1e-5)
            np.testing.assert_allclose(-0.5 * N_(kl)[0] / (B * Z), vo.kl_loss(mu, lv_ref), rtol=1e-5)
        dmu, dlv = torch.zeros(B, Z, device=dev), torch.zeros(B, Z, device=dev)
        ckl = 0.7 / (B * Z)
        ops.latent_bwd(T_(dz, dev), T_(mu, dev), T_(lv_ref, dev), T_(lvr, dev), T_(eps, dev), B, Z, sp, ckl, dmu, dlv)
        np.testing.assert_allclose(N_(dmu), dz + ckl * mu, atol=1e-6)
        ref = dz * eps * 0.5 * np.exp(0.5 * lv_ref) + 0.5 * ckl * (np.exp(lv_ref) - 1)
        if sp:
            ref = ref * vo.sigmoid(lvr)
        np.testing.assert_allclose(N_(dlv), ref, atol=1e-5)


def philox4x32_10(ctr, key):
    """numpy Philox4x32-10 (Salmon et al., Random123): ctr (n, 4) uint32, key (2,) uint32 -> (n, 4) uint32.  Test infrastructure: pinned below
    against the published known-answer vectors, then used to pin the device draw of vame_latent_fwd_f32."""
    c = np.array(ctr, dtype=np.uint64, copy=True)
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    M0, M1, m32 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xffffffff)
    for _ in range(10):
        p0, p1 = M0 * c[:, 0], M1 * c[:, 2]
        n0 = (p1 >> np.uint64(32)) ^ c[:, 1] ^ k0
        n2 = (p0 >> np.uint64(32)) ^ c[:, 3] ^ k1
        c = np.stack([n0, p1 & m32, n2, p0 & m32], axis=1)
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & m32, (k1 + np.uint64(0xBB67AE85)) & m32
    return c.astype(np.uint32)


def philox_normal_ref(idx, seed, step):
    """float64 restatement of the kernel's draw: Box-Muller on words 0, 1 of Philox(counter = (idx, step), key = seed)."""
    idx = np.asarray(idx, dtype=np.uint64)
    ctr = np.stack([idx & np.uint64(0xffffffff), idx >> np.uint64(32), np.full_like(idx, step & 0xffffffff), np.full_like(idx, step >> 32)], axis=1)
    r = philox4x32_10(ctr, (seed & 0xffffffff, seed >> 32)).astype(np.float64)
    u1 = (np.floor(r[:, 0] / 256) + 0.5) / 16777216.0
    u2 = (np.floor(r[:, 1] / 256) + 0.5) / 16777216.0
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def check_latent_draw(dev, n_rows=33334):
    """The reparameterisation's N(0,1) draw made inside vame_latent_fwd_f32 (rng != NULL; the reference's torch.randn_like, rnn_model.py:71-74):
    the generator against its published known-answer vectors, the kernel's values against the float64 restatement, N(0,1) moments and a
    Kolmogorov-Smirnov test on ~1e6 samples, reproducibility per (seed, step), independence of the launch size, and the device-side
    step counter (consecutive launches draw fresh values with identical arguments)."""
    from scipy import stats
    kat = philox4x32_10([[0, 0, 0, 0], [0xffffffff] * 4, [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344]], (0, 0))[0]
    assert [hex(v) for v in kat] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    assert [hex(v) for v in philox4x32_10([[0xffffffff] * 4], (0xffffffff, 0xffffffff))[0]] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    assert [hex(v) for v in philox4x32_10([[0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344]], (0xa4093822, 0x299f31d0))[0]] == \
        ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]
    Z = 30
    seed = 0x1234567811223344

    def draw(B, st, launches=1):
        rng = torch.tensor([seed, st, 0, 0], dtype=torch.int64, device=dev)
        mu, lv = torch.zeros(B, Z, device=dev), torch.zeros(B, Z, device=dev)
        eps, z, lvo = torch.empty(B, Z, device=dev), torch.empty(B, Z, device=dev), torch.empty(B, Z, device=dev)
        outs = []
        for _ in range(launches):
            ops.latent_fwd(mu, lv, eps, B, Z, 0, 1, lvo, z, None, rng=rng)
            outs.append(N_(eps).copy())
            np.testing.assert_array_equal(N_(z), outs[-1])                 # mu = 0, logvar = 0: z = eps
        return outs, N_(rng)
    (e0, e1), state = draw(n_rows, 5, launches=2)
    assert list(state) == [seed, 7, 0, 0]                                   # two launches advanced the step twice, ticket back at 0
    x = e0.ravel().astype(np.float64)
    assert not np.array_equal(e0, e1)
    np.testing.assert_allclose(x[:4096], philox_normal_ref(np.arange(4096), seed, 5), atol=2e-5)
    np.testing.assert_allclose(e1.ravel()[-4096:], philox_normal_ref(np.arange(x.size - 4096, x.size), seed, 6), atol=2e-5)
    n = x.size
    assert abs(x.mean()) < 4.0 / np.sqrt(n) and abs(x.var() - 1.0) < 4.0 * np.sqrt(2.0 / n)
    assert abs(stats.skew(x)) < 4.0 * np.sqrt(6.0 / n) and abs(stats.kurtosis(x)) < 4.0 * np.sqrt(24.0 / n)
    assert stats.kstest(x, "norm").statistic < 1.95 / np.sqrt(n)                # alpha = 0.001
    assert np.abs(x).max() < 6.0
    # same (seed, step) -> same bits, whatever the launch size; another step / seed -> another draw
    (r0,), _ = draw(n_rows, 5)
    np.testing.assert_array_equal(r0, e0)
    (small,), _ = draw(100, 5)
    np.testing.assert_array_equal(small.ravel(), e0.ravel()[:3000])
    assert abs(np.corrcoef(e0.ravel(), e1.ravel())[0, 1]) < 5.0 / np.sqrt(n)


def check_loss_finish(dev):
    """vame_loss_finish_f32: scaled terms, weighted total, float64 epoch accumulators, sums zeroed (rnn_vae.py:129-150)."""
    raw0 = np.array([3.5e4, 1.7e4, -812.25, 2.75, 9.0, 9.0, 9.0, 9.0], np.float32)
    scale, wts = (1.0, 0.5, -0.5 / 900, 1.0), (1.0, 1.0, 0.35, 0.7)
    acc = torch.zeros(6, dtype=torch.float64, device=dev)
    tot = []
    for with_fut in (1, 0):
        raw, out = T_(raw0, dev), torch.zeros(5, device=dev)
        ops.loss_finish(raw, scale, wts, with_fut, out, acc)
        t = raw0[:4] * np.array(scale, np.float32)
        if not with_fut:
            t[1] = 0
        np.testing.assert_allclose(N_(out)[:4], t, rtol=1e-6)
        np.testing.assert_allclose(N_(out)[4], float(np.dot(t.astype(np.float64), wts)), rtol=1e-6)
        assert not N_(raw).any()
        tot.append(N_(out).astype(np.float64))
    a = N_(acc)
    np.testing.assert_allclose(a[0], tot[0][4] + tot[1][4], rtol=1e-12)
    np.testing.assert_allclose(a[1:5], tot[0][:4] + tot[1][:4], rtol=1e-12)
    assert a[5] == tot[1][4]
    out = torch.zeros(5, device=dev)
    ops.loss_finish(T_(raw0, dev), scale, wts, 1, out, None)                 # no accumulator


def check_mse(dev):
    rng = np.random.default_rng(4)
    B, TF, row = 11, 30 * 24, 45 * 24
    pred = rng.standard_normal((B, TF)).astype(np.float32)
    win = rng.standard_normal((B, row)).astype(np.float32)
    dp, loss = torch.zeros(B, TF, device=dev), torch.zeros(2, device=dev)
    ops.mse_fwd_bwd(T_(pred, dev), T_(win, dev), 24, row, B, TF, 2.0, dp, loss, 1)
    tgt = win[:, 24:24 + TF]
    np.testing.assert_allclose(N_(loss)[1], ((pred - tgt) ** 2).sum(), rtol=1e-5)
    np.testing.assert_allclose(N_(dp), 2 * (pred - tgt), atol=1e-6)
    # unaligned rows (scalar path), more rows than workgroups, no dpred
    B, TF, row = 2500, 9 * 10, 13 * 10 + 3
    pred = rng.standard_normal((B, TF)).astype(np.float32)
    win = rng.standard_normal((B, row)).astype(np.float32)
    loss = torch.zeros(2, device=dev)
    ops.mse_fwd_bwd(T_(pred, dev), T_(win, dev), 7, row, B, TF, 1.0, None, loss, 0)
    np.testing.assert_allclose(N_(loss)[0], ((pred - win[:, 7:7 + TF]) ** 2).sum(), rtol=2e-5)


def check_head_fused(dev, B=7, T=9, F=24, K=64, pad=2):
    """Streaming output head (Linear -> MSE(sum) -> dpred -> dY, and the Linear's weight gradient, one pass over the states) vs float64 numpy, on the
    decoder-state layout (B, T+pad, K) with a row offset, a ragged last 16-row tile, a target window inside a wider row, dY rows wider than K and
    dW at an offset inside a larger gradient bucket."""
    rng = np.random.default_rng(21)
    Y = rng.standard_normal((B, T + pad, K)).astype(np.float32)
    W = (rng.standard_normal((F, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(F).astype(np.float32)
    row, off = (T + 5) * F + 4, 8
    win = rng.standard_normal((B, row)).astype(np.float32)
    gs = 1.7
    Yt, Wt = T_(Y, dev), T_(W, dev)
    pred, dpred = torch.zeros(B * T, F, device=dev), torch.zeros(B * T, F, device=dev)
    dY = torch.full((B * T, K + 8), 7.0, device=dev)
    loss = torch.full((3,), 0.5, device=dev)
    bucket = torch.full((F * K + 24,), 3.0, device=dev)
    n_ws = ops.head_stream_ws_floats(B * T, F, K)
    assert n_ws > 0 and ops.head_stream_ws_floats(B * T, 33, K) == -1 and ops.head_stream_ws_floats(B * T, F, K + 32) == -1 and ops.head_stream_ws_floats(B * T, F, 1152) == -1
    ws = torch.full((n_ws,), float("nan"), device=dev)              # (every word the reduction reads must have been written by the kernel)
    o1 = 1 if pad else 0
    for _ in range(2):                                              # (a second call on the same buffers: dW is overwritten, the loss slot adds)
        ops.head_stream(Operand(Yt, K, off=o1 * K, seg=T, seg_stride=(T + pad) * K), B * T, F, K, Operand(Wt, K), T_(bias, dev), T_(win, dev), off, row, gs,
                        pred, dpred, dY, K + 8, loss, 1, bucket, 16, ws)
    y = Y[:, o1:o1 + T].reshape(B * T, K).astype(np.float64)
    p = y @ W.astype(np.float64).T + bias
    tgt = win[:, off:off + T * F].reshape(B * T, F)
    e = p - tgt
    np.testing.assert_allclose(N_(pred), p, atol=2e-5)
    np.testing.assert_allclose(N_(dpred), gs * e, atol=5e-5)
    np.testing.assert_allclose(N_(loss)[1], 0.5 + 2 * (e ** 2).sum(), rtol=2e-5)
    assert N_(loss)[0] == 0.5 and N_(loss)[2] == 0.5
    out = N_(dY)
    np.testing.assert_allclose(out[:, :K], (gs * e) @ W.astype(np.float64), atol=1e-4)
    assert (out[:, K:] == 7.0).all()
    dW_ref = (gs * e).T @ y
    got = N_(bucket)
    np.testing.assert_allclose(got[16:16 + F * K].reshape(F, K), dW_ref, atol=3e-5 * max(1.0, float(np.abs(dW_ref).max())))
    assert (got[:16] == 3.0).all() and (got[16 + F * K:] == 3.0).all()


def check_linear_group(dev):
    """vame_linear_group_f32: several Linear layers of one narrow (K <= 32) input in one launch vs float64 numpy -- ragged row and column tiles,
    problems wider than one 256-column tile, a missing bias, an output with a wider row stride, odd K and N (scalar-store path)."""
    rng = np.random.default_rng(44)
    for (M, K, Ns) in ((37, 30, (768, 512, 768)), (300, 7, (33, 256, 257, 5)), (64, 32, (96,) * 8), (5, 1, (3,))):
        A = rng.standard_normal((M, K + 3)).astype(np.float32)
        At = T_(A, dev)
        probs, refs = [], []
        for i, N in enumerate(Ns):
            W = rng.standard_normal((N, K)).astype(np.float32)
            b = rng.standard_normal(N).astype(np.float32) if i != 1 else None
            ldc = N + (4 if i == 0 else 0)
            C = torch.full((M, ldc), 9.0, device=dev)
            probs.append((Operand(T_(W, dev), K), T_(b, dev) if b is not None else None, C, ldc, N))
            refs.append(A[:, :K].astype(np.float64) @ W.astype(np.float64).T + (b if b is not None else 0.0))
        ops.linear_group(Operand(At, K + 3), M, K, probs)
        for (W, b, C, ldc, N), ref in zip(probs, refs):
            out = N_(C)
            np.testing.assert_allclose(out[:, :N], ref, atol=3e-5)
            assert (out[:, N:] == 9.0).all()


def check_colsum(dev):
    rng = np.random.default_rng(6)
    a = rng.standard_normal((13, 300)).astype(np.float32)
    out = torch.ones(260, device=dev)
    ops.colsum(T_(a, dev), 5, 13, 260, 300, out, 0, accumulate=True)
    np.testing.assert_allclose(N_(out), 1 + a[:, 5:265].sum(0), atol=1e-5)
    for (R, C) in ((5000, 24), (4100, 64), (4096, 12)):        # narrow contiguous matrices: the float4-stream kernel
        b = rng.standard_normal((R, C)).astype(np.float32)
        o = torch.zeros(C, device=dev)
        ops.colsum(T_(b, dev), 0, R, C, C, o, 0)
        np.testing.assert_allclose(N_(o), b.astype(np.float64).sum(0), atol=2e-3)
    # wide aligned matrices with >= 1024 rows: the 16-byte form (all of a thread's loads in flight); a column offset inside wider rows, ragged last
    # slab, a column count that is not a multiple of 256, accumulate
    for (R, C, ld, off) in ((4096, 512, 512, 0), (1030, 260, 520, 8), (9001, 768, 1024, 256)):
        b = rng.standard_normal((R, ld)).astype(np.float32)
        o = torch.full((C + 4,), 2.0, device=dev)
        ops.colsum(T_(b, dev), off, R, C, ld, o, 0, accumulate=True)
        got = N_(o)
        np.testing.assert_allclose(got[:C], 2.0 + b[:, off:off + C].astype(np.float64).sum(0), atol=3e-3)
        assert (got[C:] == 2.0).all()


def check_colsum_batch(dev):
    rng = np.random.default_rng(16)
    a = rng.standard_normal((128, 1024)).astype(np.float32)
    b = rng.standard_normal((37, 30)).astype(np.float32)
    at, bt = T_(a, dev), T_(b, dev)
    out = torch.zeros(2000, device=dev)
    ops.colsum_batch([(at, 0, 128, 768, 1024, out, 0), (at, 768, 128, 256, 1024, out, 800), (bt, 0, 37, 30, 30, out, 1100)])
    o = N_(out)
    np.testing.assert_allclose(o[:768], a[:, :768].sum(0), atol=1e-4)
    np.testing.assert_allclose(o[800:1056], a[:, 768:].sum(0), atol=1e-4)
    np.testing.assert_allclose(o[1100:1130], b.sum(0), atol=1e-5)
    assert (o[768:800] == 0).all() and (o[1130:] == 0).all()


def check_adam(dev):
    rng = np.random.default_rng(7)
    n = 1000
    p0 = rng.standard_normal(n).astype(np.float32)
    pt = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.Adam([pt], lr=5e-4, amsgrad=True)
    p, m, v, vm = T_(p0, dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    p2, m2, v2, vm2 = T_(p0, dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    state = torch.zeros(4, dtype=torch.int32, device=dev)
    state.view(torch.float32)[0] = 5e-4
    for step in range(1, 4):
        g = rng.standard_normal(n).astype(np.float32) * (10.0 ** (step - 2))
        pt.grad = torch.from_numpy(g.copy())
        opt.step()
        ops.adam_amsgrad(p, T_(2 * g, dev), m, v, vm, n, 5e-4, step, gscale=0.5)
        np.testing.assert_allclose(N_(p), pt.detach().numpy(), atol=1e-6)
        # the same update with the learning rate and the step number read from (and counted on) the device: identical arguments every step
        ops.adam_amsgrad(p2, T_(2 * g, dev), m2, v2, vm2, n, 123.0, 0, gscale=0.5, state=state)
        np.testing.assert_allclose(N_(p2), pt.detach().numpy(), atol=1e-6)
        assert N_(state)[1:3].tolist() == [step, 0]


def check_nuclear(dev, B, Z, k):
    rng = np.random.default_rng(8)
    z = (rng.standard_normal((B, Z)) * rng.uniform(0.2, 2.0, Z)).astype(np.float32)
    G = (z.astype(np.float64).T @ z.astype(np.float64)).astype(np.float32)
    loss, Minv = torch.zeros(1, device=dev), torch.zeros(Z, Z, device=dev)
    ops.nuclear(T_(G, dev), Z, k, B, 0.1, B, loss, 0, Minv)
    ref_loss, ref_dz = vo.cluster_loss_gram(z, k, 0.1, B)
    assert abs(N_(loss)[0] - ref_loss) <= 1e-5 * abs(ref_loss)
    assert abs(N_(loss)[0] - vo.cluster_loss_svd(z, k, 0.1, B)) <= 1e-4 * abs(ref_loss)
    np.testing.assert_allclose(z @ N_(Minv), ref_dz, atol=2e-5 * np.abs(ref_dz).max())
    # warm start: a second, slightly perturbed Gram solved from the stored eigenvectors gives the same answer as a cold solve
    vst = torch.zeros(ops.nuclear_state_doubles(Z), device=dev, dtype=torch.float64)
    ops.nuclear(T_(G, dev), Z, k, B, 0.1, B, loss, 0, Minv, vstate=vst)
    z2 = (z + 0.01 * rng.standard_normal(z.shape)).astype(np.float32)
    G2 = (z2.astype(np.float64).T @ z2.astype(np.float64)).astype(np.float32)
    ops.nuclear(T_(G2, dev), Z, k, B, 0.1, B, loss, 0, Minv, vstate=vst)
    ref_loss2, ref_dz2 = vo.cluster_loss_gram(z2, k, 0.1, B)
    assert abs(N_(loss)[0] - ref_loss2) <= 1e-5 * abs(ref_loss2)
    np.testing.assert_allclose(z2 @ N_(Minv), ref_dz2, atol=2e-5 * np.abs(ref_dz2).max())


def check_kmeans(dev, N=3000, K=6, D=30, n_init=3):
    """GPU k-means vs scikit-learn: same partition on separable data, inertia within rounding; the E-step kernel
    vs a numpy argmin."""
    from sklearn.cluster import KMeans
    from sklearn.metrics import adjusted_rand_score
    from vame_amd.analysis.kmeans_hip import KMeansHIP
    rng = np.random.default_rng(0)
    cent = rng.standard_normal((K, D)) * 4
    X = (cent[rng.integers(0, K, N)] + rng.standard_normal((N, D))).astype(np.float32)
    C = X[rng.choice(N, K, replace=False)]
    labels = torch.empty(N, dtype=torch.int32, device=dev)
    mind2 = torch.empty(N, device=dev)
    Kp = (K + 3) // 4 * 4
    onehot = torch.empty(N, Kp, device=dev)
    ops.kmeans_assign(T_(X, dev), N, D, T_(C, dev), K, labels, mind2, onehot, Kp)
    d2 = ((X[:, None, :].astype(np.float64) - C[None].astype(np.float64)) ** 2).sum(-1)
    np.testing.assert_array_equal(N_(labels), d2.argmin(1))
    np.testing.assert_allclose(N_(mind2), d2.min(1), rtol=1e-5)
    oh = N_(onehot)
    assert (oh.sum(1) == 1).all() and (oh.argmax(1) == d2.argmin(1)).all()
    km = KMeansHIP(K, n_init=n_init, random_state=42).fit(X)
    sk = KMeans(init="k-means++", n_clusters=K, random_state=42, n_init=n_init).fit(X)
    assert abs(km.inertia_ - sk.inertia_) <= 1e-5 * sk.inertia_
    assert adjusted_rand_score(sk.labels_, km.labels_) == 1.0
    assert (km.predict(X) == km.labels_).all()
    order = np.argsort(km.cluster_centers_[:, 0])
    order_sk = np.argsort(sk.cluster_centers_[:, 0])
    np.testing.assert_allclose(km.cluster_centers_[order], sk.cluster_centers_[order_sk], atol=1e-4)


# ------------------------------------------------------------------ training-set preparation (float64)
def _prep_kw(g):
    f, L, p, _ = g["params"]
    return dict(robust=True, iqr_factor=int(f), savgol_filter=True, savgol_length=int(L), savgol_order=int(p))


def check_prepare_series_golden(dev):
    """Device pipeline of vame_amd.model.create_training against the REFERENCE's outputs (tests/golden/prep_*.npz): bit-exact."""
    from conftest import load_golden
    from vame_amd.model.create_training import prepare_series
    for name, fixed in (("prep_aligned", False), ("prep_fixed", True)):
        g = load_golden(name)
        out, pos, info = prepare_series([g["in0"], g["in1"]], fixed=fixed, device=torch.device(dev), **_prep_kw(g))
        ntest = g["test"].shape[1]
        np.testing.assert_array_equal(out[:, :ntest], g["test"])
        np.testing.assert_array_equal(out[:, ntest:], g["train"])
        np.testing.assert_array_equal(out[:, pos[0]:pos[1]], g["clean0"])
        np.testing.assert_array_equal(out[:, pos[1]:pos[2]], g["clean1"])
        assert (info["anchors"] == (7, 3)) if not fixed else info["anchors"] is None


def check_prepare_series_vs_oracle(dev, F=26, sizes=(5000, 3001), seed=5, L=7, p=3):
    """Other sizes / filter settings against the numpy restatement (oracle/prep_oracle.py), incl. no-savgol and not-robust."""
    from oracle import prep_oracle as po
    from vame_amd.model.create_training import prepare_series
    rng = np.random.default_rng(seed)
    datas = []
    for N in sizes:
        X = rng.standard_normal((F, N)).cumsum(axis=1) * 0.1 + rng.standard_normal((F, N))
        idx = rng.integers(0, N, size=N // 25)
        X[rng.integers(0, F, size=len(idx)), idx] *= 40.0
        X[1] = 0.0
        X[F - 2] = 0.0
        datas.append(X)
    for fixed in (False, True):
        for robust, sg in ((True, True), (True, False), (False, True)):
            kw = dict(fixed=fixed, robust=robust, iqr_factor=3, savgol_filter=sg, savgol_length=L, savgol_order=p)
            ref = po.traindata(datas, test_fraction=0.1, **kw)
            out, pos, _ = prepare_series(datas, device=torch.device(dev), **kw)
            np.testing.assert_array_equal(out, np.concatenate([ref["test"], ref["train"]], axis=1), err_msg=str(kw))


def check_prep_fill_rules(dev):
    """The two NaN-fill kernels against np.interp called exactly as the reference's interpol() calls it, incl. runs of NaNs at
    the ends, a feature without any valid sample (aligned rule) and the empty-frame count (fixed rule)."""
    from vame_amd import ops
    rng = np.random.default_rng(11)
    F, N = 9, 257
    z = rng.standard_normal((F, N))
    z[rng.random((F, N)) < 0.2] = np.nan
    z[2, :40] = np.nan
    z[5, -30:] = np.nan
    z[:, 100] = np.nan                                   # one frame with no valid feature
    z[0, 7] = np.nan
    z[F - 1, 9] = np.nan
    # --- fixed rule, frame by frame (create_training.py:236)
    ref = z.T.copy()
    xs = np.arange(F, dtype=float)
    for i in range(N):
        nan = np.isnan(ref[i])
        if nan.any() and not nan.all():
            ref[i, nan] = np.interp(xs[nan], xs[~nan], ref[i, ~nan])
    t = torch.from_numpy(z.copy()).to(dev)
    n_empty = torch.zeros(1, dtype=torch.int32, device=dev)
    ops.prep_fill_across_features(t, F, N, N, n_empty)
    np.testing.assert_array_equal(t.cpu().numpy(), ref.T)
    assert int(n_empty.item()) == 1
    # --- aligned rule on the whole array (create_training.py:27-32): np.interp over the FEATURE index
    za = z.copy()
    za[:, 100] = 0.5
    y = za.copy()
    nans = np.isnan(y)
    y[nans] = np.interp(nans.nonzero()[0], (~nans).nonzero()[0], y[~nans])
    t = torch.from_numpy(za.copy()).to(dev)
    fl = torch.empty(F, 2, dtype=torch.float64, device=dev)
    ws = ops.prep_ws(F, t.device)
    ops.prep_fill_last_valid(t, F, N, N, fl, ws)
    np.testing.assert_array_equal(t.cpu().numpy(), y)
    # a feature without valid samples is reported (NaN, NaN) and left alone; the host rule reproduces np.interp for it
    from vame_amd.model.create_training import _resolve_empty_features
    for empty in (0, 4, F - 1):
        zb = za.copy()
        zb[empty] = np.nan
        y = zb.copy()
        nans = np.isnan(y)
        y[nans] = np.interp(nans.nonzero()[0], (~nans).nonzero()[0], y[~nans])
        t = torch.from_numpy(zb.copy()).to(dev)
        ops.prep_fill_last_valid(t, F, N, N, fl, ws)
        assert np.isnan(fl.cpu().numpy()[empty]).all()
        _resolve_empty_features(t, fl)
        np.testing.assert_array_equal(t.cpu().numpy(), y)


def check_adam_abort_and_mask_scale(dev):
    """vame_adam_amsgrad_f32 leaves everything untouched when its abort word is set; vame_mask_scale_f32 (dropout) on the padded
    sequence layout and in place."""
    rng = np.random.default_rng(17)
    n = 999
    p0 = rng.standard_normal(n).astype(np.float32)
    p, g = T_(p0, dev), T_(rng.standard_normal(n).astype(np.float32), dev)
    m, v, vm = torch.zeros(n, device=dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    flag = torch.ones(1, dtype=torch.int32, device=dev)
    ops.adam_amsgrad(p, g, m, v, vm, n, 5e-4, 1, abort_flag=flag)
    np.testing.assert_array_equal(N_(p), p0)
    assert float(m.abs().sum()) == 0 and float(v.abs().sum()) == 0 and float(vm.abs().sum()) == 0
    flag.zero_()
    ops.adam_amsgrad(p, g, m, v, vm, n, 5e-4, 1, abort_flag=flag)
    assert np.abs(N_(p) - p0).max() > 1e-5
    B, T, C = 5, 7, 24
    Y = rng.standard_normal((B, T + 2, C)).astype(np.float32)
    mask = (rng.random((B, T, C)) > 0.25).astype(np.float32)
    out = torch.empty(B, T, C, device=dev)
    ops.mask_scale(T_(Y, dev), C, C, T, (T + 2) * C, T_(mask, dev), 1.0 / 0.75, out, B * T, C)
    np.testing.assert_allclose(N_(out), Y[:, 1:T + 1] * mask * np.float32(1.0 / 0.75), rtol=1e-6)
    d = T_(Y[:, 1:T + 1].copy(), dev)
    ops.mask_scale(d, 0, C, 0, 0, T_(mask, dev), 2.0, d, B * T, C)
    np.testing.assert_allclose(N_(d), Y[:, 1:T + 1] * mask * 2.0, rtol=1e-6)


def _hmm_data(rng, N, K, D):
    """A sticky K-state Gaussian chain: states are recoverable but overlapping enough for soft posteriors."""
    A = np.full((K, K), 0.08 / (K - 1))
    np.fill_diagonal(A, 0.92)
    means = rng.standard_normal((K, D)) * 2.0
    scales = rng.uniform(0.5, 1.5, (K, D))
    z = np.empty(N, dtype=int)
    z[0] = rng.integers(K)
    for t in range(1, N):
        z[t] = rng.choice(K, p=A[z[t - 1]])
    X = means[z] + rng.standard_normal((N, D)) * scales[z]
    return X.astype(np.float32), z


def check_hmm(dev, N=700, K=4, D=6, chunk=64):
    """GaussianHMMHIP (vame_amd/analysis/hmm_hip.py, float64 kernels) against the numpy restatement of hmmlearn's algorithm
    (oracle/hmm_oracle.py): one E-step (log-likelihood, every sufficient statistic), the whole EM trajectory and the Viterbi path."""
    from oracle.hmm_oracle import GaussianHMMOracle, log_mvn_density_full
    from vame_amd.analysis.hmm_hip import GaussianHMMHIP
    rng = np.random.default_rng(11)
    X, z = _hmm_data(rng, N, K, D)
    means0 = X[rng.choice(N, K, replace=False)].astype(np.float64)
    ref = GaussianHMMOracle(K, n_iter=6)
    ref.init_params(X.astype(np.float64), means0)
    # perturbed parameters (non-uniform start / transition probabilities) for the single E-step comparison
    ref.startprob_ = rng.dirichlet(np.ones(K))
    ref.transmat_ = rng.dirichlet(np.ones(K) * 2, size=K)
    hip = GaussianHMMHIP(K, n_iter=6, chunk=chunk)
    hip._init_params(X, means0)
    hip.startprob_, hip.transmat_ = ref.startprob_.copy(), ref.transmat_.copy()
    Xd = hip._upload(X)
    b = hip._buffers(N, D, Xd.device)
    ll, st = hip._e_step(Xd, b)
    ll_ref, st_ref, post_ref = ref.e_step(X.astype(np.float64))
    np.testing.assert_allclose(b["logB"].cpu().numpy().reshape(N, K), log_mvn_density_full(X.astype(np.float64), ref.means_, ref.covars_), rtol=1e-9, atol=1e-9)
    assert abs(ll - ll_ref) <= 1e-8 * abs(ll_ref), (ll, ll_ref)
    np.testing.assert_allclose(b["gamma"].cpu().numpy().reshape(N, K), post_ref, atol=1e-10)
    for k in ("post", "start", "trans", "obs", "obsobs"):
        np.testing.assert_allclose(st[k], st_ref[k], rtol=1e-6, atol=1e-9 * max(1.0, np.abs(st_ref[k]).max()), err_msg=k)   # (the log-domain oracle cancels ~1e6-sized terms)
    assert abs(st["trans"].sum() - (N - 1)) < 1e-6 and abs(st["post"].sum() - N) < 1e-6
    # EM trajectory from the same initial means
    ref2 = GaussianHMMOracle(K, n_iter=6).fit(X.astype(np.float64), means0)
    hip2 = GaussianHMMHIP(K, n_iter=6, chunk=chunk).fit(X, means=means0)
    np.testing.assert_allclose(hip2.history_, ref2.history, rtol=1e-8)
    assert all(b2 >= a2 - 1e-6 for a2, b2 in zip(hip2.history_, hip2.history_[1:]))           # EM never decreases the likelihood
    np.testing.assert_allclose(hip2.means_, ref2.means_, atol=1e-7)
    np.testing.assert_allclose(hip2.covars_, ref2.covars_, atol=1e-7)
    np.testing.assert_allclose(hip2.transmat_, ref2.transmat_, atol=1e-8)
    # Viterbi: same path and score; the path's score is the maximum (>= the true states' joint log-probability)
    lp, path = hip2.decode(X)
    from oracle.hmm_oracle import log_mask_zero, viterbi_log
    logB = log_mvn_density_full(X.astype(np.float64), ref2.means_, ref2.covars_)
    lp_ref, path_ref = viterbi_log(log_mask_zero(ref2.startprob_), log_mask_zero(ref2.transmat_), logB)
    np.testing.assert_array_equal(path, path_ref)
    assert abs(lp - lp_ref) <= 1e-8 * abs(lp_ref)
    np.testing.assert_array_equal(hip2.predict(X), path_ref)
    import pickle
    clone = pickle.loads(pickle.dumps(hip2))                                                   # results/hmm_trained.pkl round trip
    np.testing.assert_array_equal(clone.predict(X), path_ref)
