#!/usr/bin/env python3
"""Timing-only ablations of the split-bf16 contraction on the headline launch (6 x 768 x 256 x 122,880).  Tuning build:
    make ab && VAME_LIB=tools/libvame_hip_ab.so python tools/split_abl.py [reps]
opt: bits 0-1 accumulators (0 = two, 1 = one); tuning build: bit 3 = the 4 + 4 wave-specialised mapping, bit 7 = 256 x 128 tiles, bit 2 = 4 MFMA + 8 split
waves (tools/gemm_split_variants.inc); bits 8.. (gemm.hip SPLIT_ABL): 1 no fragment reads / MFMAs, 2 reads but no MFMAs, 4 no split / LDS stores,
8 no global loads, 16 split but no LDS stores, 32 the loads as dwordx4."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vame_amd import _lib, ops  # noqa: E402
from vame_amd.ops import Operand  # noqa: E402

if os.environ.get("VAME_LIB"):
    _lib._lib = _lib._bind(os.environ["VAME_LIB"])
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n, M, N, K, sk = 6, 768, 256, 4096 * 30, 32
PAD = int(os.environ.get("PAD", "0"))          # extra floats per operand row: 0 = the engine's power-of-two pitches (4 KB / 2 KB ... here 1 KB for B)
SK = int(os.environ.get("SK", "32"))
sk = SK
A = [torch.randn(K, M + 256 + PAD, device="cuda") for _ in range(n)]
B = [torch.randn(K, 2 * N + PAD, device="cuda") for _ in range(n)]       # pitch 2H like the (B, T + 2, 2H) sequences
C = torch.empty(n * M * N, device="cuda")
ws = torch.empty(n * sk * M * N, device="cuda")
opA, opB = [Operand(a, M + 256 + PAD) for a in A], [Operand(b, 2 * N + PAD) for b in B]
print(f"row pad {PAD} floats, split-K {sk}")


def t(split):
    f = lambda: ops.gemm_group(M, N, K, opA, 1, opB, 1, C, [g * M * N for g in range(n)], N, sk, ws, a_gap_at=512, a_gap=256, split=split)  # noqa: E731
    f()
    torch.cuda.synchronize()
    probe = ops.ClockProbe(torch.device("cuda", 0))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    probe.start()
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    probe.stop()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, probe.mhz()


rows = [("f32 MFMA kernel (vame_gemm_group_f32)", None),
        ("split: symmetric waves, 2 accumulators, 2 WG/CU", 0), ("split: symmetric waves, 1 accumulator, 3 WG/CU", 1),
        ("  no MFMA (reads kept)", 1 | (2 << 8)), ("  no reads, no MFMA", 1 | (1 << 8)), ("  no split / store", 1 | (4 << 8)), ("  no global loads", 1 | (8 << 8)),
        ("  loads only", 1 | (5 << 8)), ("  loads only as dwordx4", 1 | (37 << 8)), ("  split + store only", 1 | (9 << 8)), ("  split VALU only", 1 | (25 << 8)),
        ("  reads + MFMA only", 1 | (12 << 8)), ("  barriers only", 1 | (13 << 8)),
        ("variant: 4 MFMA + 4 split waves, 3 LDS images, 2 acc", 8), ("  consumers idle (producers alone)", 8 | (1 << 8)), ("  producers load only, consumers idle", 8 | (5 << 8)),
        ("  consumers alone", 8 | (12 << 8)),
        ("variant: 256 x 128 tiles, 8 waves, 2 images, 2 acc", 128),
        ("variant: 256 x 128, hand-ordered stream (1 acc)", 128 | 16 | 1), ("  loads only", 128 | (5 << 8)), ("  split + store only", 128 | (9 << 8)), ("  reads + MFMA only", 128 | (12 << 8)),
        ("variant: 4 MFMA + 8 split waves, 3 images, 1 acc", 4), ("  consumers idle (producers alone)", 4 | (1 << 8)), ("  loads only, consumers idle", 4 | (5 << 8)),
        ("  split + store only", 4 | (9 << 8)), ("  consumers alone", 4 | (12 << 8))]
ref = None
for label, opt in rows:
    try:
        ms, mhz = t(opt)
        if opt is None:
            ref = C.clone()
        elif (opt >> 8) == 0 and ref is not None:          # a complete (un-ablated) split form: its result against the f32-input kernel's
            err = float((C - ref).abs().max() / ref.abs().max())
            label = f"{label} [max diff vs f32 kernel {err:.1e}]"
        print(f"{label:60s} {ms:8.3f} ms  {2.0 * M * N * K * n / ms / 1e9:8.1f} TF   shader clock {mhz or 0:6.0f} MHz")
    except Exception as e:               # the product build refuses the ablation bits
        print(f"{label:60s} refused: {str(e)[:80]}")
