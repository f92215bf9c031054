#!/usr/bin/env python3
"""Per-kernel sums of rocprofv3 --pmc counter CSVs (one pass per counter set, as the gfx950 slot
limits require) -> HBM traffic per launch with the gfx950 FETCH_SIZE correction
(MI355X_MICROARCH.md section HBM: FETCH_SIZE reads exactly 1/2 of a wide coalesced stream; unit KiB)."""
import csv
import json
import re
import sys
from collections import defaultdict


def load(path):
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    with open(path) as f:
        for r in csv.DictReader(f):
            name = re.sub(r"\(.*\)$", "", r["Kernel_Name"]).replace("tmk::", "").replace("void ", "")
            agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[name].add(r["Dispatch_Id"])
    return agg, {k: len(v) for k, v in calls.items()}


def main(fetch_csv, write_csv, sq_csv=None, out_json=None, prefix="conv3d_mfma<2,"):
    """prefix: the dominant kernel whose launches are averaged into `hbm_bytes_per_launch` (bench.py's
    `roofline.traffic`): 'conv3d_mfma<2,' for the fp32 path, 'conv27_bf16' for the bf16 path."""
    if sq_csv in ("", "-"):
        sq_csv = None
    f, nf = load(fetch_csv)
    w, _ = load(write_csv)
    sq, _ = load(sq_csv) if sq_csv else ({}, {})
    rows = []
    for k in sorted(f, key=lambda k: -f[k].get("FETCH_SIZE", 0)):
        fs, ws = f[k].get("FETCH_SIZE", 0.0), w.get(k, {}).get("WRITE_SIZE", 0.0)
        n = nf[k]
        rd, wr = 2.0 * fs * 1024, ws * 1024
        row = {"kernel": k, "launches": n, "hbm_read_bytes_per_launch": rd / n, "hbm_write_bytes_per_launch": wr / n,
               "hbm_bytes_per_launch": (rd + wr) / n}
        if k in sq and sq[k].get("SQ_BUSY_CYCLES"):
            row["mfma_busy_over_sq_busy"] = sq[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / sq[k]["SQ_BUSY_CYCLES"]
        rows.append(row)
    print(f"{'kernel':28s} {'launches':>8s} {'rd MB/launch':>13s} {'wr MB/launch':>13s} {'mfma_busy/sq_busy':>18s}")
    for r in rows:
        print(f"{r['kernel'][:28]:28s} {r['launches']:8d} {r['hbm_read_bytes_per_launch'] / 1e6:13.2f} "
              f"{r['hbm_write_bytes_per_launch'] / 1e6:13.2f} {r.get('mfma_busy_over_sq_busy', float('nan')):18.3f}")
    conv = [r for r in rows if r["kernel"].startswith(prefix)]
    n = sum(r["launches"] for r in conv)
    tot = sum(r["hbm_bytes_per_launch"] * r["launches"] for r in conv)
    summary = {"kernel": prefix + "*> (3x3x3 conv, all tile instantiations)", "launches": n, "hbm_bytes_per_launch": tot / max(1, n),
               "note": "FETCH_SIZE doubled per the gfx950 correction; separate --pmc passes", "per_kernel": rows}
    print(f"{prefix}*: {n} launches, {tot / max(1, n) / 1e6:.1f} MB HBM traffic per launch")
    if out_json:
        json.dump(summary, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:])
