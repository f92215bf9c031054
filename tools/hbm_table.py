#!/usr/bin/env python3
"""Per-kernel HBM table of one bench workload: joins the kernel-trace durations (rocprofv3 --kernel-trace .db) with the PMC
traffic of the same workload (tools/pmc_summary.py JSON: FETCH_SIZE / WRITE_SIZE passes, gfx950 correction applied) ->
launches, average duration, HBM bytes per launch, achieved GB/s and the fraction of the 8 TB/s HBM3E roof, for every kernel
above a share of the step's kernel time.  MFMA kernels carry their matrix-pipe fraction beside it (they are not HBM-bound).
usage: hbm_table.py trace.db traffic.json [min_pct]"""
import json
import re
import sqlite3
import sys

HBM_PEAK = 8.0e12


def short(name):
    return re.sub(r"\(.*\)$", "", name).replace("tmk::", "").replace("void ", "")


def main(db_path, traffic_json, min_pct="0.5"):
    cur = sqlite3.connect(db_path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else cols[0]
    agg = {}
    for n, s, e in cur.execute(f"select {namecol}, start, end from kernels order by start"):
        a = agg.setdefault(short(n), [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e9
    tot = sum(a[1] for a in agg.values())
    tr = {r["kernel"]: r for r in json.load(open(traffic_json))["per_kernel"]}
    print(f"{'kernel':46s} {'calls':>6s} {'avg_us':>9s} {'pct':>6s} {'rd MB':>9s} {'wr MB':>9s} {'GB/s':>8s} {'of 8 TB/s':>9s} {'mfma x clk/2.4':>14s}")
    for n, (calls, secs) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        pct = 100 * secs / tot
        if pct < float(min_pct):
            continue
        r = tr.get(n)
        avg = secs / calls
        if r is None:
            print(f"{n[:46]:46s} {calls:6d} {avg * 1e6:9.1f} {pct:6.2f}   (no counter row)")
            continue
        gbs = r["hbm_bytes_per_launch"] / avg
        mf = r.get("frac_of_mfma_peak")
        print(f"{n[:46]:46s} {calls:6d} {avg * 1e6:9.1f} {pct:6.2f} {r['hbm_read_bytes_per_launch'] / 1e6:9.1f} "
              f"{r['hbm_write_bytes_per_launch'] / 1e6:9.1f} {gbs / 1e9:8.0f} {gbs / HBM_PEAK:9.3f} "
              f"{(f'{mf:14.3f}' if mf else ' ' * 14)}")


if __name__ == "__main__":
    main(*sys.argv[1:])
