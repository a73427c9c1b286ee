#!/usr/bin/env python3
"""Stress the cross-workgroup hand-offs of the cooperative GRU kernels: many launches on the SAME buffers with fresh data each
time (a stale L1 / L2 line or a stale hand-off packet would show), a second stream streaming memory in the background (uneven load), every
word checked: against the batch-tile-persistent kernels to summation-order rounding, and BIT FOR BIT between the two forms of the
cooperative launch (16-row groups with tagged packets on twice the workgroups vs 32-row groups), which exchange different packets in a
different order.  usage: python tools/coop_stress.py [iterations]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

from vame_amd import ops
from vame_amd.ops import GB, GF
from kernel_cases import _gru_weights, _pack, _valid_stash_mask

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda")
H, T = 256, 30
rng = np.random.default_rng(0)
state = ops.CoopState(dev)
side = torch.cuda.Stream()
junk_a, junk_b = torch.empty(64 << 20, device=dev), torch.empty(64 << 20, device=dev)
bad = 0
for B in (256, 100, 512):
    nt = (B + 31) // 32
    W = [_gru_weights(rng, 8, H) for _ in range(2)]
    packs = [_pack("cuda", w[1], w[2], w[3], H) for w in W]
    gi = [torch.empty(B, T, 3 * H, device=dev) for _ in range(2)]
    h0 = torch.empty(B, H, device=dev)
    Y = [torch.zeros(B, T + 2, 2 * H, device=dev) for _ in range(3)]
    hN = [torch.zeros(B, 2 * H, device=dev) for _ in range(3)]
    stash = [[torch.zeros(ops.gru_stash_floats(B, T, H), device=dev) for _ in range(2)] for _ in range(3)]
    dY, dhN = torch.empty(B, T, 2 * H, device=dev), torch.empty(B, 2 * H, device=dev)
    outs = [[(torch.zeros(B, T, 4 * H, device=dev), torch.zeros(B, H, device=dev), torch.zeros(nt, 4 * H, device=dev)) for _ in range(2)]
            for _ in range(3)]

    smask = torch.from_numpy(_valid_stash_mask(B, T, H).copy()).to(dev)        # stash entries of rows past the batch are don't-cares

    def rows_f(k):
        return [{GF["GI"]: ops.addr(gi[d]), GF["GI_ROW"]: T * 3 * H, GF["GI_T"]: 3 * H, GF["WP"]: ops.addr(packs[d][0]),
                 GF["BHN"]: ops.addr(packs[d][3]), GF["H0"]: ops.addr(h0) if d else 0, GF["H0_ROW"]: H,
                 GF["Y"]: ops.addr(Y[k], 2 * H + d * H), GF["Y_ROW"]: (T + 2) * 2 * H, GF["Y_T"]: 2 * H,
                 GF["HN"]: ops.addr(hN[k], d * H), GF["HN_ROW"]: 2 * H, GF["STASH"]: ops.addr(stash[k][d]), GF["T"]: T,
                 GF["REVERSE"]: d, GF["PAD"]: 1} for d in range(2)]

    def rows_b(k):
        return [{GB["STASH"]: ops.addr(stash[0][d]), GB["Y"]: ops.addr(Y[0], 2 * H + d * H), GB["Y_ROW"]: (T + 2) * 2 * H,
                 GB["Y_T"]: 2 * H, GB["WPT"]: ops.addr(packs[d][1]), GB["DY"]: ops.addr(dY, d * H), GB["DY_ROW"]: T * 2 * H,
                 GB["DY_T"]: 2 * H, GB["DHN"]: ops.addr(dhN, d * H), GB["DHN_ROW"]: 2 * H, GB["DG"]: ops.addr(outs[k][d][0]),
                 GB["DH0"]: ops.addr(outs[k][d][1]), GB["DH0_ROW"]: H, GB["DBIAS"]: ops.addr(outs[k][d][2]),
                 GB["T"]: T, GB["REVERSE"]: d, GB["PAD"]: 1} for d in range(2)]

    for it in range(iters):
        for g_ in gi:
            g_.normal_()
        h0.normal_(0, 0.5)
        dY.normal_()
        dhN.normal_()
        with torch.cuda.stream(side):                       # background traffic on other CUs / the same memory system
            for _ in range(3):
                junk_b.copy_(junk_a)
        ops.gru_seq_fwd(rows_f(0), B, H)
        ops.gru_coop_fwd(rows_f(1), B, H, state)
        ops.gru_coop_fwd(rows_f(2), B, H, state, kernel=ops.KERNEL_LOCKSTEP)
        ops.gru_seq_bwd(rows_b(0), B, H)
        ops.gru_coop_bwd(rows_b(1), B, H, state)
        ops.gru_coop_bwd(rows_b(2), B, H, state, kernel=ops.KERNEL_LOCKSTEP)
        torch.cuda.synchronize()
        ok = bool((Y[0] - Y[1]).abs().max() <= 2e-6) and bool((hN[0] - hN[1]).abs().max() <= 2e-6)
        ok = ok and torch.equal(Y[1], Y[2]) and torch.equal(hN[1], hN[2])                   # the two cooperative forms: the same bits
        for d in range(2):
            ok = ok and torch.equal(stash[1][d][smask], stash[2][d][smask])
            for a, b, c in zip(outs[0][d][:2], outs[1][d][:2], outs[2][d][:2]):
                tol = 2e-5 * max(1.0, float(a.abs().max()))
                ok = ok and bool((a - b).abs().max() <= tol) and torch.equal(b, c)
            ok = ok and torch.equal(outs[1][d][2], outs[2][d][2])                              # bias partials incl. the 16-row hand-over
        if not ok:
            bad += 1
            print(f"MISMATCH B={B} iteration {it}", flush=True)
    print(f"B={B}: {iters} iterations checked, mismatches so far {bad}, poll timeouts {int(state.status.item())}", flush=True)
print("STRESS", "FAILED" if bad or int(state.status.item()) else "OK")
