#!/usr/bin/env python3
"""Kernel micro-benchmarks on the MI355X (HIP events on the launch stream): the GEMM shapes and GRU
sequence launches of the B=4096 train step.  Usage: python tools/microbench.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402

from vame_amd import _lib, ops  # noqa: E402
if os.environ.get("VAME_LIB"):          # A/B against another build of the library
    _lib._lib = _lib._bind(os.environ["VAME_LIB"])
from vame_amd.ops import GB, GF, Operand  # noqa: E402

dev = "cuda"
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def timeit(fn, reps=REPS):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def bench_gemm(M, N, K, akm, bkm, sk=1, label=""):
    A = torch.randn((K, M) if akm else (M, K), device=dev)
    B = torch.randn((K, N) if bkm else (N, K), device=dev)
    C = torch.empty(M, N, device=dev)
    ws = torch.empty(sk * M * N, device=dev) if sk > 1 else None
    ms = timeit(lambda: ops.gemm(M, N, K, Operand(A, A.shape[1]), akm, Operand(B, B.shape[1]), bkm, C, N, splitk=sk, ws=ws))
    print(f"gemm {label:10s} M={M:6d} N={N:4d} K={K:6d} akm={akm} bkm={bkm} sk={sk:3d}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TF")


def bench_gru(H, B, T, nstreams, quiet=False, hook=None):
    from kernel_cases import _gru_weights, _pack
    import numpy as np
    rng = np.random.default_rng(0)
    rows_f, rows_b, keep = [], [], []
    ntiles = (B + 31) // 32
    for s in range(nstreams):
        W_ih, W_hh, b_ih, b_hh = _gru_weights(rng, 8, H)
        wpf, wpb, bgi, bhn = _pack(dev, W_hh, b_ih, b_hh, H)
        gi = torch.randn(B, T, 3 * H, device=dev)
        Y = torch.zeros(B, T + 2, 2 * H, device=dev)
        stash = torch.zeros(ops.gru_stash_floats(B, T, H), device=dev)
        dG = torch.zeros(B, T, 4 * H, device=dev)
        dY = torch.randn(B, T, 2 * H, device=dev)
        dbias = torch.zeros(ntiles, 4 * H, device=dev)
        d = s % 2
        rows_f.append({GF["GI"]: ops.addr(gi), GF["GI_ROW"]: T * 3 * H, GF["GI_T"]: 3 * H, GF["WP"]: ops.addr(wpf),
                       GF["BHN"]: ops.addr(bhn), GF["Y"]: ops.addr(Y, 2 * H + d * H), GF["Y_ROW"]: (T + 2) * 2 * H,
                       GF["Y_T"]: 2 * H, GF["STASH"]: ops.addr(stash), GF["T"]: T, GF["REVERSE"]: d, GF["PAD"]: 1})
        rows_b.append({GB["STASH"]: ops.addr(stash), GB["Y"]: ops.addr(Y, 2 * H + d * H), GB["Y_ROW"]: (T + 2) * 2 * H,
                       GB["Y_T"]: 2 * H, GB["WPT"]: ops.addr(wpb), GB["DY"]: ops.addr(dY, d * H), GB["DY_ROW"]: T * 2 * H,
                       GB["DY_T"]: 2 * H, GB["DG"]: ops.addr(dG), GB["DBIAS"]: ops.addr(dbias), GB["T"]: T, GB["REVERSE"]: d,
                       GB["PAD"]: 1})
        keep.append((wpf, wpb, bgi, bhn, gi, Y, stash, dG, dY, dbias))
    fl = nstreams * 2.0 * 3 * H * H * B * T
    ms = timeit(lambda: ops.gru_seq_fwd(rows_f, B, H))
    if not quiet:
        print(f"gru_fwd H={H} B={B} T={T} streams={nstreams}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF")
    if hook:
        hook("fwd")
    ms2 = timeit(lambda: ops.gru_seq_bwd(rows_b, B, H))
    if hook:
        hook("bwd")
    if not quiet:
        print(f"gru_bwd H={H} B={B} T={T} streams={nstreams}: {ms2*1e3:8.1f} us  {fl/ms2/1e9:7.1f} TF")
    if H % 64 == 0 and H >= 128 and not quiet:        # the two-blocks-per-wave kernels on the same launch
        ms3 = timeit(lambda: ops.gru_wide_fwd(rows_f, B, H))
        ms4 = timeit(lambda: ops.gru_wide_bwd(rows_b, B, H))
        print(f"wide_fwd H={H} B={B} T={T} streams={nstreams}: {ms3*1e3:8.1f} us  {fl/ms3/1e9:7.1f} TF")
        print(f"wide_bwd H={H} B={B} T={T} streams={nstreams}: {ms4*1e3:8.1f} us  {fl/ms4/1e9:7.1f} TF")
    return ms * 1e3, ms2 * 1e3


def ablate():
    """GRU kernel ablations; needs the tuning build: make ab && VAME_LIB=tools/libvame_hip_ab.so python tools/microbench.py 10 ablate"""
    for var, masks in (("VAME_ABL_FWD", (0, 512, 2, 0, 512, 2)), ("VAME_ABL_BWD", (0, 256, 1024, 0, 256, 1024))):
        for m in masks:
            os.environ[var] = str(m)
            print(f"--- {var}={m}")
            bench_gru(256, 4096, 30, 2)
        os.environ[var] = "0"


def gemm_tiles(rounds=4):
    """Interleaved A/B of GEMM tile configurations (library built with `make ab`): 0 = 128x128/4 waves, 1 = 256x128/4 waves
    (128x64 per wave), 2 = 256x128/8 waves."""
    import statistics
    BT = 4096 * 30
    shapes = [(BT, 768, 512, 0, 0, 1), (BT, 512, 768, 0, 1, 1), (768, 512, BT, 1, 1, 32), (768, 256, BT, 1, 1, 64), (768, 256, BT, 1, 1, 32),
              (4096, 4096, 4096, 1, 1, 1)]
    for (M, N, K, akm, bkm, sk) in shapes:
        A = torch.randn((K, M) if akm else (M, K), device=dev)
        B = torch.randn((K, N) if bkm else (N, K), device=dev)
        C = torch.empty(M, N, device=dev)
        ws = torch.empty(sk * M * N, device=dev) if sk > 1 else None
        res = {v: [] for v in (0, 1, 2)}
        for r in range(rounds):
            for v in res:
                os.environ["VAME_GEMM_TILE"] = str(v)
                ms = timeit(lambda: ops.gemm(M, N, K, Operand(A, A.shape[1]), akm, Operand(B, B.shape[1]), bkm, C, N, splitk=sk, ws=ws), reps=5)
                res[v].append(2.0 * M * N * K / ms / 1e9)
        print(f"M={M} N={N} K={K} akm={akm} bkm={bkm} sk={sk}: " + "  ".join(f"tile{v}: {statistics.median(t):6.1f}" for v, t in res.items()))


def gemm_skinny(rounds=4):
    """Tile configurations for the N <= 32 shapes of the step (tuning build): 0 = 128x32 / 4 waves (product), 3 = 64x32 / 2 waves,
    4 = 256x32 / 8 waves."""
    import statistics
    BT = 4096 * 30
    shapes = [(BT, 24, 512, 0, 0, 1), (BT // 2, 24, 512, 0, 0, 1), (768, 24, BT, 1, 1, 128), (4096, 30, 768, 0, 1, 1), (4096, 30, 1024, 0, 0, 1),
              (768, 30, 4096, 1, 1, 8)]
    for (M, N, K, akm, bkm, sk) in shapes:
        A = torch.randn((K, M) if akm else (M, K), device=dev)
        B = torch.randn((K, N) if bkm else (N, K), device=dev)
        C = torch.empty(M, N, device=dev)
        ws = torch.empty(sk * M * N, device=dev) if sk > 1 else None
        res = {v: [] for v in (0, 3, 4, 10, 11, 15)}            # 10 / 11 / 15: the 128x32 tile with loop variant 0 / 1 / 5
        for r in range(rounds):
            for v in res:
                os.environ["VAME_GEMM_TILE"] = str(v if v < 10 else 0)
                os.environ["VAME_GEMM_VAR_SKINNY"] = str(v - 10) if v >= 10 else "-1"
                ms = timeit(lambda: ops.gemm(M, N, K, Operand(A, A.shape[1]), akm, Operand(B, B.shape[1]), bkm, C, N, splitk=sk, ws=ws), reps=10)
                res[v].append(ms * 1e3)
        print(f"M={M} N={N} K={K} akm={akm} bkm={bkm} sk={sk}: " + "  ".join(f"tile{v}: {statistics.median(t):7.1f} us" for v, t in res.items()), flush=True)


def gemm_ab(rounds=4):
    """Interleaved A/B of the GEMM kernel variants (library built with `make ab`)."""
    BT = 4096 * 30
    shapes = [(BT, 768, 512, 0, 0, 1), (BT, 512, 768, 0, 1, 1), (768, 512, BT, 1, 1, 32), (768, 256, BT, 1, 1, 64), (4096, 4096, 4096, 1, 1, 1)]
    if len(sys.argv) > 3 and sys.argv[3] == "tinyk":      # the K = 24 / 30 forms (output-head dY, projections of z): epilogue-bound
        shapes = [(BT, 512, 24, 0, 1, 1), (BT // 2, 512, 24, 0, 1, 1), (4096, 768, 30, 0, 0, 1), (4096, 512, 30, 0, 0, 1), (BT, 768, 24, 0, 0, 1)]
    import statistics
    for (M, N, K, akm, bkm, sk) in shapes:
        A = torch.randn((K, M) if akm else (M, K), device=dev)
        B = torch.randn((K, N) if bkm else (N, K), device=dev)
        C = torch.empty(M, N, device=dev)
        ws = torch.empty(sk * M * N, device=dev) if sk > 1 else None
        res = {v: [] for v in (0, 1, 5, 9, 13)}
        for r in range(rounds):
            for v in res:
                os.environ["VAME_GEMM_VAR"] = str(v)
                ms = timeit(lambda: ops.gemm(M, N, K, Operand(A, A.shape[1]), akm, Operand(B, B.shape[1]), bkm, C, N, splitk=sk, ws=ws), reps=5)
                res[v].append(2.0 * M * N * K / ms / 1e9)
        print(f"M={M} N={N} K={K} akm={akm} bkm={bkm} sk={sk}: " + "  ".join(f"v{v}: {statistics.median(t):6.1f} (max {max(t):6.1f})" for v, t in res.items()))


def gemm_sk(rounds=3):
    """split-K sweep of the weight-gradient (TN) shapes for the plain (VAR 5, 3 workgroups per CU) and the software-pipelined
    (VAR 13, 2 per CU) loops, single and grouped launches (tuning build)."""
    import statistics
    BT = 4096 * 30
    for (M, N, K, cnt, sks) in [(768, 512, BT, 1, (16, 21, 32, 42, 64)), (768, 256, BT, 1, (32, 42, 64, 85, 128)), (768, 256, BT, 6, (8, 14, 16, 21, 32)),
                                (768, 512, BT, 2, (8, 16, 21, 32)), (768, 256, BT // 2, 2, (16, 21, 32, 42))]:
        As = [torch.randn(K, M, device=dev) for _ in range(cnt)]
        Bs = [torch.randn(K, N, device=dev) for _ in range(cnt)]
        C = torch.empty(cnt * M, N, device=dev)
        for sk in sks:
            ws = torch.empty(cnt * sk * M * N, device=dev)
            res = {}
            for r in range(rounds):
                for v in (5, 13):
                    os.environ["VAME_GEMM_VAR"] = str(v)
                    if cnt == 1:
                        fn = lambda: ops.gemm(M, N, K, Operand(As[0], M), 1, Operand(Bs[0], N), 1, C, N, splitk=sk, ws=ws)
                    else:
                        fn = lambda: ops.gemm_group(M, N, K, [Operand(a, M) for a in As], 1, [Operand(b, N) for b in Bs], 1, C, [g * M * N for g in range(cnt)], N, sk, ws)
                    ms = timeit(fn, reps=5)
                    res.setdefault(v, []).append(2.0 * M * N * K * cnt / ms / 1e9)
            print(f"M={M} N={N} K={K} x{cnt} sk={sk:3d}: " + "  ".join(f"v{v}: {statistics.median(t):6.1f}" for v, t in res.items()), flush=True)


def gemm_epi(rounds=4):
    """Interleaved A/B of the GEMM epilogue forms (library built with `make ab`; VAME_GEMM_EPI): 0 = 16 dword stores per 32x32 tile
    (lane = column), 1 = swapped MFMA operands, 4 dwordx4 stores (lane = row), 2 = LDS-transposed, 4 dwordx4 full-line stores."""
    import statistics
    BT = 4096 * 30
    shapes = [(BT, 768, 512, 0, 0, 1), (BT, 512, 768, 0, 1, 1), (BT, 768, 24, 0, 0, 1), (BT, 24, 512, 0, 0, 1), (768, 512, BT, 1, 1, 32), (768, 256, BT, 1, 1, 64),
              (8192, 1536, 512, 0, 0, 1), (4096, 4096, 4096, 0, 0, 1)]
    for (M, N, K, akm, bkm, sk) in shapes:
        A = torch.randn((K, M) if akm else (M, K), device=dev)
        B = torch.randn((K, N) if bkm else (N, K), device=dev)
        C = torch.empty(M, N, device=dev)
        ws = torch.empty(sk * M * N, device=dev) if sk > 1 else None
        res = {v: [] for v in (0, 1, 2)}
        for r in range(rounds):
            for v in res:
                os.environ["VAME_GEMM_EPI"] = str(v)
                ms = timeit(lambda: ops.gemm(M, N, K, Operand(A, A.shape[1]), akm, Operand(B, B.shape[1]), bkm, C, N, splitk=sk, ws=ws), reps=5)
                res[v].append(2.0 * M * N * K / ms / 1e9)
        print(f"M={M} N={N} K={K} akm={akm} bkm={bkm} sk={sk}: " + "  ".join(f"epi{v}: {statistics.median(t):6.1f} (max {max(t):6.1f})" for v, t in res.items()), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "gemm_sk":
        gemm_sk()
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "gemm_epi":
        gemm_epi()
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "gemm_tiles":
        gemm_tiles()
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "gemm_skinny":
        gemm_skinny()
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "gemm_ab":
        gemm_ab()
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "gru":
        for (h, b, t, n) in ((256, 4096, 30, 2), (256, 4096, 30, 4), (256, 4096, 15, 2), (128, 4096, 30, 2), (512, 8192, 60, 2)):
            bench_gru(h, b, t, n)
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "ablate":
        ablate()
        sys.exit(0)
    BT = 4096 * 30
    bench_gemm(BT, 768, 512, 0, 0, label="gi_L1")
    bench_gemm(BT, 768, 24, 0, 0, label="gi_L0")
    bench_gemm(BT, 512, 768, 0, 1, label="dY0")
    bench_gemm(BT, 512, 24, 0, 1, label="dYdec")
    for sk in (16, 32, 64):
        bench_gemm(768, 512, BT, 1, 1, sk, label="dWih_L1")
    for sk in (32, 64, 128):
        bench_gemm(512, 256, BT, 1, 1, sk, label="dWhh_a")
    bench_gemm(256, 256, BT, 1, 1, 64, label="dWhh_b")
    bench_gemm(768, 24, BT, 1, 1, 128, label="dWih_L0")
    bench_gemm(4096, 4096, 4096, 0, 0, label="square")
    bench_gemm(4096, 4096, 4096, 1, 1, label="squareTN")
    bench_gru(256, 4096, 30, 2)
    bench_gru(256, 4096, 30, 4)
    bench_gru(256, 8192, 30, 2)
