"""Forward GRU sequence kernels on the MI355X: lock-step vs skewed (two wave groups half a step apart) per hidden size and stream
form, interleaved and warmed, HIP-event medians.  Forms: `gi` = per-step gi (encoder layer 1), `xin` = fused input projection (encoder
layer 0), `dec` = four streams with a time-constant gi and an initial state, 2 x T + 2 x T/2 steps (decoder + future decoder).
usage: python tools/fwd_table.py [B] [T]"""
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from vame_amd import ops
from vame_amd.ops import GF
dev = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 30
SIZES = [int(v) for v in os.environ.get("FWD_H", "256,192,128,64").split(",")]
DELAYS = [int(v) for v in os.environ.get("FWD_DELAYS", "0,8,16,24,32").split(",")]


NOSTASH = os.environ.get("FWD_NOSTASH", "0") != "0"      # inference form (no BPTT stash): traffic experiments
FORMS = os.environ.get("FWD_FORMS", "").split(",") if os.environ.get("FWD_FORMS") else None


def rows_for(form, H):
    rows, flops, keep = _rows_for(form, H)
    if NOSTASH:
        for r in rows: r[GF["STASH"]] = 0
    if os.environ.get("FWD_NOY", "0") != "0":                # no output sequence either (only the final state would leave: traffic experiments)
        for r in rows: r[GF["Y"]] = 0; r[GF["PAD"]] = 0
    if os.environ.get("FWD_NSTREAMS"):
        rows = rows[:int(os.environ["FWD_NSTREAMS"])]
    return rows, flops, keep


def _rows_for(form, H):
    keep, rows = [], []
    F = 24
    def pack():
        W = torch.randn(3 * H, H, device=dev) / H ** 0.5
        b1, b2 = torch.randn(3 * H, device=dev) * 0.1, torch.randn(3 * H, device=dev) * 0.1
        wpf, wpb, bgi, bhn = (torch.empty(3 * H * H, device=dev), torch.empty(3 * H * H, device=dev), torch.empty(3 * H, device=dev), torch.empty(H, device=dev))
        ops.gru_pack(W, b1, b2, H, wpf, wpb, bgi, bhn)
        keep.extend([W, b1, b2, wpf, wpb, bgi, bhn])
        return wpf, bgi, bhn
    if form in ("gi", "xin"):
        Y, hN = torch.zeros(B, T + 2, 2 * H, device=dev), torch.zeros(B, 2 * H, device=dev)
        win = torch.randn(B, T + 15, F, device=dev)
        keep.extend([Y, hN, win])
        for d in range(2):
            wpf, bgi, bhn = pack()
            st = torch.empty(ops.gru_stash_floats(B, T, H), device=dev)
            keep.append(st)
            r = {GF["WP"]: ops.addr(wpf), GF["BHN"]: ops.addr(bhn), GF["H0"]: 0, GF["H0_ROW"]: H, GF["Y"]: ops.addr(Y, 2 * H + d * H),
                 GF["Y_ROW"]: (T + 2) * 2 * H, GF["Y_T"]: 2 * H, GF["HN"]: ops.addr(hN, d * H), GF["HN_ROW"]: 2 * H, GF["STASH"]: ops.addr(st),
                 GF["T"]: T, GF["REVERSE"]: d, GF["PAD"]: 1}
            if form == "gi":
                gi = torch.randn(B, T, 3 * H, device=dev)
                keep.append(gi)
                r.update({GF["GI"]: ops.addr(gi), GF["GI_ROW"]: T * 3 * H, GF["GI_T"]: 3 * H})
            else:
                Wi = torch.randn(3 * H, F, device=dev) * 0.2
                wpx = torch.zeros(3 * H * 32, device=dev)
                ops.gru_pack_x(Wi, F, H, wpx)
                keep.extend([Wi, wpx])
                r.update({GF["GI"]: ops.addr(win), GF["GI_ROW"]: (T + 15) * F, GF["GI_T"]: F, GF["WPX"]: ops.addr(wpx), GF["BGI"]: ops.addr(bgi), GF["XF"]: F})
            rows.append(r)
        flops = 2 * 2.0 * 3 * H * H * B * T
    else:
        flops = 0.0
        for steps in (T, T // 2):
            Y = torch.zeros(B, steps + 2, 2 * H, device=dev)
            hid = torch.randn(2 * B, H, device=dev) * 0.5
            keep.extend([Y, hid])
            for d in range(2):
                wpf, bgi, bhn = pack()
                st = torch.empty(ops.gru_stash_floats(B, steps, H), device=dev)
                gi = torch.randn(B, 3 * H, device=dev)
                keep.extend([st, gi])
                rows.append({GF["GI"]: ops.addr(gi), GF["GI_ROW"]: 3 * H, GF["GI_T"]: 0, GF["WP"]: ops.addr(wpf), GF["BHN"]: ops.addr(bhn),
                             GF["H0"]: ops.addr(hid, d * B * H), GF["H0_ROW"]: H, GF["Y"]: ops.addr(Y, 2 * H + d * H), GF["Y_ROW"]: (steps + 2) * 2 * H,
                             GF["Y_T"]: 2 * H, GF["HN"]: 0, GF["HN_ROW"]: 0, GF["STASH"]: ops.addr(st), GF["T"]: steps, GF["REVERSE"]: d, GF["PAD"]: 1})
                flops += 2.0 * 3 * H * H * B * steps
        rows = [rows[0], rows[1], rows[2], rows[3]]
    return rows, flops, keep


def timed(rows, H, kernel, prio, n=3, delay=-1):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if H > 256:
            for i in range(0, len(rows), 2): ops.gru_wide_fwd(rows[i:i + 2], B, H, kernel=kernel)
        else:
            ops.gru_seq_fwd(rows, B, H, kernel=kernel, prio=prio, delay=delay)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return ts


def main():
  print(f"B={B} T={T}; us per launch (median of 18, interleaved), TF = algorithmic flops / time")
  for H in SIZES:
      for form in (FORMS or (("gi", "xin", "dec") if H <= 256 else ("gi", "dec"))):
          rows, flops, keep = rows_for(form, H)
          variants = [("lock-step", ops.KERNEL_LOCKSTEP, -1)]
          if H > 256:
              variants += [("skewed", ops.KERNEL_SKEWED, -1)] if H % 128 == 0 else []
          elif ops.gru_seq_fwd_has_kernel(H, ops.KERNEL_SKEWED):
              variants += [(f"skew d{d}", ops.KERNEL_SKEWED, d) for d in DELAYS]
          for _ in range(4):
              for _, k, d in variants: timed(rows, H, k, 0, 2, delay=d)
          res = {name: [] for name, _, _ in variants}
          for _ in range(6):
              for name, k, d in variants: res[name] += timed(rows, H, k, 0, delay=d)
          line = f"H={H:3d} {form:3s}: "
          for name in res:
              v = sorted(res[name]); med = v[len(v) // 2]
              line += f"{name} {med:7.1f} us {flops / med / 1e6:5.1f} TF | "
          print(line, flush=True)
          del rows, keep
          torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
