"""BPTT kernels on the MI355X per hidden size: lock-step vs wave-specialised (pacing sweep), interleaved and warmed, HIP-event medians,
plus a bit-for-bit check of dG / dh0 between the two.  usage: python tools/bwd_table.py [B] [T]"""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import kernel_cases as kc
from vame_amd import ops
dev = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 30
SIZES = [int(v) for v in os.environ.get("BWD_H", "256,192,128,64").split(",")]
print(f"B={B} T={T}, 2 streams; us per launch (median of 15, interleaved)")
for H in SIZES:
    x, st, Y, hN = kc.run_gru_fwd(dev, H, B, T, seed=1)
    dY = torch.randn(B, T, 2 * H, device=dev); dhN = torch.randn(B, 2 * H, device=dev)
    rows, outs = [], []
    ntiles = (B + 31) // 32
    GB = ops.GB
    for d, s in enumerate(st):
        dG = torch.zeros(B, T, 4 * H, device=dev); dh0 = torch.zeros(B, H, device=dev); dbias = torch.zeros(ntiles, 4 * H, device=dev)
        rows.append({GB["STASH"]: ops.addr(s["stash"]), GB["Y"]: ops.addr(Y, 2 * H + d * H), GB["Y_ROW"]: (T + 2) * 2 * H, GB["Y_T"]: 2 * H,
                     GB["WPT"]: ops.addr(s["wpb"]), GB["DY"]: ops.addr(dY, d * H), GB["DY_ROW"]: T * 2 * H, GB["DY_T"]: 2 * H,
                     GB["DHN"]: ops.addr(dhN, d * H), GB["DHN_ROW"]: 2 * H, GB["DG"]: ops.addr(dG), GB["DH0"]: ops.addr(dh0), GB["DH0_ROW"]: H,
                     GB["DBIAS"]: ops.addr(dbias), GB["T"]: T, GB["REVERSE"]: d, GB["PAD"]: 1})
        outs.append((dG, dh0))
    base = 2 * H * H // 65536, 8 * H * H // 65536
    variants = [("lock-step", ops.KERNEL_LOCKSTEP, -1, -1)]
    if ops.gru_seq_bwd_has_kernel(H, ops.KERNEL_WS):
        pairs = sorted({base, (0, 0), (max(base[0] // 2, 0), max(base[1] // 2, 1)), (base[0] + 1, base[1] + 2)})
        if os.environ.get("BWD_PACE"):                      # explicit sweep: BWD_PACE="1/8,2/6,2/7"
            pairs = [tuple(int(v) for v in pr.split("/")) for pr in os.environ["BWD_PACE"].split(",")]
        variants += [(f"ws {cp}/{ld}", ops.KERNEL_WS, cp, ld) for cp, ld in pairs]
    def run(k, cp, ld, n=3):
        ts = []
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.gru_seq_bwd(rows, B, H, kernel=k, pace_cp=cp, pace_ld=ld); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        return ts
    ref = None
    same = True
    for name, k, cp, ld in variants:
        run(k, cp, ld, 1)
        cur = [(a.clone(), b.clone()) for a, b in outs]
        if ref is None: ref = cur
        else: same = same and all(torch.equal(a, c) and torch.equal(b, d) for (a, b), (c, d) in zip(cur, ref))
    for _ in range(3):
        for _, k, cp, ld in variants: run(k, cp, ld, 2)
    res = {v[0]: [] for v in variants}
    for _ in range(5):
        for name, k, cp, ld in variants: res[name] += run(k, cp, ld)
    fl = 2 * 2.0 * 3 * H * H * B * T
    line = f"H={H:3d}: "
    for name in res:
        v = sorted(res[name]); med = v[len(v) // 2]
        line += f"{name} {med:7.1f} us {fl / med / 1e6:5.1f} TF | "
    print(line + f"dG/dh0 bit-identical: {same}", flush=True)
    del rows, outs, x, st, Y, hN
    torch.cuda.empty_cache()
