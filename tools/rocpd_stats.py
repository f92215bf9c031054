#!/usr/bin/env python3
"""Summarise a rocprofv3 results .db (rocpd sqlite) into a per-kernel stats table
(name, calls, total ms, avg us, % of GPU kernel time) -- the `--stats` summary in text form."""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*\)$", "", name)
    name = name.replace("tmk::", "")
    return name[:70]


def main(path, skip_first=0):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else cols[0]
    rows = cur.execute(f"select {namecol}, start, end from kernels order by start").fetchall()
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(short(n), [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print(f"# {path}: {len(rows)} dispatches, {tot / 1e3:.3f} ms total kernel time")
    print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{n:70s} {a[0]:7d} {a[1] / 1e3:10.3f} {a[1] / a[0]:10.1f} {a[2]:9.1f} {a[3]:9.1f} {100 * a[1] / tot:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
