#!/usr/bin/env python3
"""Column-split (cooperative) GRU forward vs the batch-tile-persistent kernel at small batches.  usage: python tools/coop_bench.py"""
import os
import sys

sys.argv = sys.argv[:1] + ["10"]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402
import microbench as mb  # noqa: E402
from vame_amd import ops  # noqa: E402

state = ops.CoopState(torch.device("cuda"))
orig, orig_b = ops.gru_seq_fwd, ops.gru_seq_bwd
for (H, B, T, ns) in ((256, 256, 30, 2), (256, 512, 30, 2), (256, 128, 30, 2), (256, 256, 30, 4), (256, 64, 30, 2)):
    if not ops.gru_coop_supported(ns, B, H):
        print(f"B={B} streams={ns}: not supported")
        continue
    f0, b0 = mb.bench_gru(H, B, T, ns, quiet=True)
    res = {}
    for name, kern in (("auto", ops.KERNEL_AUTO), ("32-row groups", ops.KERNEL_LOCKSTEP)):
        ops.gru_seq_fwd = lambda rows, B_, H_, *a, **k: ops.gru_coop_fwd(rows, B_, H_, state, kernel=kern)
        ops.gru_seq_bwd = lambda rows, B_, H_, *a, **k: ops.gru_coop_bwd(rows, B_, H_, state)
        res[name] = mb.bench_gru(H, B, T, ns, quiet=True)
        ops.gru_seq_fwd, ops.gru_seq_bwd = orig, orig_b
    f1, b1 = res["auto"]
    f2, b2 = res["32-row groups"]
    print(f"H={H} B={B} T={T} streams={ns}: fwd persistent {f0:7.1f} us, cooperative {f1:7.1f} us ({f1/T:4.1f}/step) x{f0/f1:.2f}, 32-row groups {f2:7.1f} us ({f2/T:4.1f}/step) | "
          f"bwd persistent {b0:7.1f} us, cooperative {b1:7.1f} us ({b1/T:4.1f}/step) x{b0/b1:.2f} | poll timeouts {int(state.status.item())}", flush=True)
