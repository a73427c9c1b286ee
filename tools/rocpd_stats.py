#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into a per-kernel stats table (like --stats CSV)."""
import sqlite3
import sys


def main(path, skip_first=0):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    agg = {}
    for name, s, e in rows[skip_first:]:
        d = agg.setdefault(name, [0, 0.0, 1e30, 0.0])
        dur = (e - s) / 1e3
        d[0] += 1; d[1] += dur; d[2] = min(d[2], dur); d[3] = max(d[3], dur)
    tot = sum(d[1] for d in agg.values())
    print(f"{'kernel':90s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for name, d in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{name[:90]:90s} {d[0]:6d} {d[1]:12.1f} {d[1]/d[0]:10.1f} {d[2]:10.1f} {d[3]:10.1f} {100*d[1]/tot:6.2f}")
    print(f"TOTAL kernel time {tot/1e3:.3f} ms over {len(rows)-skip_first} dispatches")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
