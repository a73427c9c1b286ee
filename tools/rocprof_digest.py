#!/usr/bin/env python3
"""Digest rocprofv3 CSV output into the compact, committed files under profiles/.

  trace  <dir> <out.csv>            per-call rows of a --kernel-trace run: short kernel name, grid (threads), workgroup size,
                                    start (us since the first dispatch), duration (us) -- so that e.g. the one 6-problem grouped
                                    GEMM launch of a step can be told apart from the other launches of the same kernel name
  stats  <dir> <out.csv>            per (kernel, grid) summary of the same run: calls, total / avg / min / max duration
  pmc    <fetch_dir> <write_dir> <lib.so> <out.json> [--key "<bench key>=><kernel substring>@<grid>"]...
                                    HBM traffic per call from two separate --pmc passes (FETCH_SIZE, WRITE_SIZE; KB units),
                                    corrected as MI355X_MICROARCH.md prescribes (gfx950 FETCH_SIZE reports 1/2 of wide coalesced
                                    reads -> x2); stores `vame_source_id()` of the library the counters were taken with (the sha256
                                    of its kernel sources) and the traffic of the launches bench.py names (`by_bench_key`) for
                                    its roofline.traffic field.
"""
import csv
import glob
import hashlib
import json
import os
import re
import sys

csv.field_size_limit(1 << 30)


def short(name):
    name = name.strip().replace("(anonymous namespace)::", "")
    m = re.match(r"^(?:void\s+)?([A-Za-z_][\w:]*)(<.*)?$", name.split("(")[0].strip())
    base = m.group(1) if m else name[:60]
    if base.startswith(("gemm_kernel", "gru_", "splitk", "nuclear", "window", "latent", "mse", "colsum", "timesum", "adam", "axpy", "mask_scale",
                        "kmeans", "prep_")):
        args = re.search(r"<([^()]*)>", name)
        return base + ("<" + args.group(1).replace(" ", "") + ">" if args else "")
    return base.split("::")[-1][:48]


def rows_of(d, suffix):
    files = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))
    if not files:
        raise SystemExit(f"no *{suffix} under {d}")
    for f in files:
        with open(f, newline="") as fh:
            yield from csv.DictReader(fh)


def trace(d, out, stats_only=False):
    rows = []
    for r in rows_of(d, "kernel_trace.csv"):
        g = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), g, int(r["Workgroup_Size_X"])))
    rows.sort()
    t0 = rows[0][0]
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        if not stats_only:
            w.writerow(["kernel", "grid_threads", "workgroup", "start_us", "duration_us"])
            for s, e, n, g, wg in rows:
                w.writerow([n, g, wg, f"{(s - t0) / 1e3:.1f}", f"{(e - s) / 1e3:.1f}"])
        else:
            agg = {}
            for s, e, n, g, wg in rows:
                a = agg.setdefault((n, g), [0, 0.0, 1e30, 0.0])
                d_ = (e - s) / 1e3
                a[0] += 1; a[1] += d_; a[2] = min(a[2], d_); a[3] = max(a[3], d_)
            tot = sum(a[1] for a in agg.values())
            w.writerow(["kernel", "grid_threads", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"])
            for (n, g), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                w.writerow([n, g, a[0], f"{a[1]:.1f}", f"{a[1] / a[0]:.1f}", f"{a[2]:.1f}", f"{a[3]:.1f}", f"{100 * a[1] / tot:.2f}"])
    print("wrote", out, len(rows), "dispatches")


def pmc(fetch_dir, write_dir, lib, out, keys, cycles=()):
    """cycles: "<kernel substring>@<grid>=label1,label2,..." -- the dispatches of that (kernel, grid) in dispatch order are labelled
    cyclically (launches of one kernel and grid that differ in what they stream, e.g. the four forward GRU launches of a step: encoder
    layer 0, layer 1, decoder, future decoder), and every label is summarised on its own as "<kernel> grid=<g> [label]"."""
    cyc = []
    for spec in cycles:
        sel, labels = spec.split("=", 1)
        sub, grid = sel.rsplit("@", 1)
        cyc.append((sub, int(grid), labels.split(",")))

    def collect(d, counter):
        rows = [r for r in rows_of(d, "counter_collection.csv") if r["Counter_Name"] == counter]
        rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0) or 0))
        agg, seen = {}, {}
        for r in rows:
            name, grid = short(r["Kernel_Name"]), int(r["Grid_Size"])
            label = ""
            for sub, g, labels in cyc:
                if sub in name and g == grid:
                    i = seen.setdefault((name, grid), {}).setdefault(r.get("Dispatch_Id"), len(seen[(name, grid)]))
                    label = " [" + labels[i % len(labels)] + "]"
            k = (name + label, grid)
            a = agg.setdefault(k, [0, 0.0])
            a[0] += 1; a[1] += float(r["Counter_Value"])
        return agg
    f, w = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, [0, 0])[1] + w.get(k, [0, 0])[1])):
        fk = f.get(k, [0, 0.0]); wk = w.get(k, [0, 0.0])
        calls = max(fk[0], wk[0])
        fetch_kb = fk[1] / max(fk[0], 1); write_kb = wk[1] / max(wk[0], 1)
        nm, lab = (k[0].split(" [")[0], " [" + k[0].split(" [")[1]) if " [" in k[0] else (k[0], "")
        kernels[f"{nm} grid={k[1]}{lab}"] = dict(calls=calls, fetch_kb_per_call=round(fetch_kb, 1), write_kb_per_call=round(write_kb, 1),
                                             hbm_bytes_per_call_corrected=int((2 * fetch_kb + write_kb) * 1024))
    import ctypes
    L = ctypes.CDLL(lib)
    L.vame_source_id.restype = ctypes.c_char_p
    sha = L.vame_source_id().decode()                              # identity of the kernel sources the profiled library was built from
    by_key = {}
    for spec in keys:
        bench_key, sel = spec.split("=>", 1)                    # "<bench key>=><kernel substring>@<grid>"
        sub, grid = sel.rsplit("@", 1)                           # "<grid>" or "<grid>[label]" (a label given by --cycle)
        lab = None
        if "[" in grid:
            grid, lab = grid.split("[", 1)
            lab = lab.rstrip("]")
        hits = [v for k, v in kernels.items() if sub in k and (k.endswith(f"grid={grid} [{lab}]") if lab else k.endswith(f"grid={grid}"))]
        if len(hits) != 1:
            raise SystemExit(f"key {bench_key!r}: {len(hits)} kernels match {sub!r} @ {grid}")
        by_key[bench_key] = dict(kernel=sub, grid_threads=int(grid), hbm_bytes_per_launch_corrected=hits[0]["hbm_bytes_per_call_corrected"],
                                 fetch_kb_per_call=hits[0]["fetch_kb_per_call"], write_kb_per_call=hits[0]["write_kb_per_call"], calls=hits[0]["calls"])
    json.dump(dict(note="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `python bench.py`; FETCH_SIZE doubled per "
                        "MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); KB = 1024 B; per call = per launch",
                   source_id=sha, by_bench_key=by_key, kernels=kernels), open(out, "w"), indent=1)
    print("wrote", out, "kernels:", len(kernels), "bench keys:", list(by_key))


def sq(d, out, note=""):
    """per (kernel, grid) means of every counter of a --pmc pass (SQ / GRBM), plus mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES /
    (GRBM_GUI_ACTIVE x 128): the share of kernel cycles the matrix pipes were busy (same normalisation as profiles/r01_pmc_gru_kernels.json)."""
    agg = {}
    for r in rows_of(d, "counter_collection.csv"):
        k = f"{short(r['Kernel_Name'])} grid={r['Grid_Size']}"
        a = agg.setdefault(k, {}).setdefault(r["Counter_Name"], [0, 0.0])
        a[0] += 1; a[1] += float(r["Counter_Value"])
    res = {}
    for k, cs in agg.items():
        e = {c: v[1] / v[0] for c, v in cs.items()}
        e["calls"] = max(v[0] for v in cs.values())
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("GRBM_GUI_ACTIVE"):
            e["mfma_busy_frac"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] * 128.0), 4)
        res[k] = e
    keep = dict(sorted(res.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0) * kv[1]["calls"])[:24])
    json.dump(dict(note=note, kernels=keep), open(out, "w"), indent=1)
    print("wrote", out, {k: v.get("mfma_busy_frac") for k, v in list(keep.items())[:8]})


if __name__ == "__main__":
    cmd = sys.argv[1]
    if cmd == "trace":
        trace(sys.argv[2], sys.argv[3])
    elif cmd == "stats":
        trace(sys.argv[2], sys.argv[3], stats_only=True)
    elif cmd == "sq":
        sq(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "")
    elif cmd == "pmc":
        keys = [sys.argv[i + 1] for i, a in enumerate(sys.argv) if a == "--key"]
        cycles = [sys.argv[i + 1] for i, a in enumerate(sys.argv) if a == "--cycle"]
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5], keys, cycles)
    else:
        raise SystemExit(__doc__)
