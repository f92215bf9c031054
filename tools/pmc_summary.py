#!/usr/bin/env python3
"""Per-kernel sums of rocprofv3 --pmc counter CSVs (one pass per counter set, as the gfx950 slot
limits require) -> HBM traffic per launch with the gfx950 FETCH_SIZE correction
(MI355X_MICROARCH.md section HBM: FETCH_SIZE reads exactly 1/2 of a wide coalesced stream; unit KiB), MFMA-busy and
the clock the chip held.

MFMA-busy normalisation (calibrated on conv3d_mfma<2,2,32>: 29.05 M v_mfma_f32_32x32x2_f32 per launch x 64 cycles =
1859 M against a counter of 1824 M): SQ_VALU_MFMA_BUSY_CYCLES is the SUM over all 1024 SIMDs (256 CUs x 4) of the cycles
their matrix pipe was busy, and GRBM_GUI_ACTIVE is the SUM over the 8 XCDs of the cycles the dispatch was active, so
    mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * GRBM_GUI_ACTIVE / 8) = SQ_VALU_MFMA_BUSY_CYCLES / (128 * GRBM_GUI_ACTIVE)
is the fraction of SIMD-cycles with the matrix pipe busy (<= 1), and
    clock_GHz = GRBM_GUI_ACTIVE / 8 / (End_Timestamp - Start_Timestamp)
is the shader clock held during the dispatch (reads high on dispatches shorter than ~0.3 ms).  The fraction of the dense
MFMA PEAK (which assumes 2.4 GHz) is mfma_busy x clock / 2.4.

usage: pmc_summary.py FETCH.csv WRITE.csv [SQ.csv|-] [out.json|-] [kernel prefix] [commit]"""
import csv
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_sha():
    """Same digest as bench.py: the sources the counters were measured on."""
    d = os.path.join(ROOT, "tera-mind_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")) or f == "Makefile":
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def load(path):
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(dict)
    with open(path) as f:
        for r in csv.DictReader(f):
            name = re.sub(r"\(.*\)$", "", r["Kernel_Name"]).replace("tmk::", "").replace("void ", "")
            agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[name][r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return agg, {k: len(v) for k, v in calls.items()}, {k: sum(v.values()) for k, v in calls.items()}


def main(fetch_csv, write_csv, sq_csv=None, out_json=None, prefix="conv3d_mfma<2,", commit=None):
    """prefix: the dominant kernel whose launches are averaged into `hbm_bytes_per_launch` (bench.py's
    `roofline.traffic`): 'conv3d_mfma<2,' for the fp32 path, 'conv27_bf16' for the bf16 path."""
    if sq_csv in ("", "-"):
        sq_csv = None
    if out_json in ("", "-"):
        out_json = None
    f, nf, _ = load(fetch_csv)
    w, _, _ = load(write_csv)
    sq, _, sq_ns = load(sq_csv) if sq_csv else ({}, {}, {})
    rows = []
    for k in sorted(f, key=lambda k: -f[k].get("FETCH_SIZE", 0)):
        fs, ws = f[k].get("FETCH_SIZE", 0.0), w.get(k, {}).get("WRITE_SIZE", 0.0)
        n = nf[k]
        rd, wr = 2.0 * fs * 1024, ws * 1024
        row = {"kernel": k, "launches": n, "hbm_read_bytes_per_launch": rd / n, "hbm_write_bytes_per_launch": wr / n,
               "hbm_bytes_per_launch": (rd + wr) / n}
        c = sq.get(k, {})
        if c.get("GRBM_GUI_ACTIVE"):
            row["mfma_busy"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (128.0 * c["GRBM_GUI_ACTIVE"])
            if sq_ns.get(k):
                row["clock_ghz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / sq_ns[k]
                row["frac_of_mfma_peak"] = row["mfma_busy"] * row["clock_ghz"] / 2.4
        rows.append(row)
    print(f"{'kernel':34s} {'launches':>8s} {'rd MB/launch':>13s} {'wr MB/launch':>13s} {'mfma_busy':>10s} {'clock GHz':>10s} {'x clk/2.4':>10s}")
    nan = float("nan")
    for r in rows:
        print(f"{r['kernel'][:34]:34s} {r['launches']:8d} {r['hbm_read_bytes_per_launch'] / 1e6:13.2f} "
              f"{r['hbm_write_bytes_per_launch'] / 1e6:13.2f} {r.get('mfma_busy', nan):10.3f} {r.get('clock_ghz', nan):10.3f} "
              f"{r.get('frac_of_mfma_peak', nan):10.3f}")
    conv = [r for r in rows if r["kernel"].startswith(prefix)]
    n = sum(r["launches"] for r in conv)
    tot = sum(r["hbm_bytes_per_launch"] * r["launches"] for r in conv)
    summary = {"kernel": prefix + "*> (3x3x3 conv, all tile instantiations)", "launches": n, "hbm_bytes_per_launch": tot / max(1, n),
               "src_sha": kernel_source_sha(), "commit": commit,
               "note": "FETCH_SIZE doubled per the gfx950 correction; separate --pmc passes; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / "
                       "(128 * GRBM_GUI_ACTIVE)", "per_kernel": rows}
    busy = [(r["mfma_busy"], r["launches"]) for r in conv if "mfma_busy" in r]
    if busy:
        summary["mfma_busy"] = sum(b * m for b, m in busy) / sum(m for _, m in busy)
        print(f"{prefix}*: launch-weighted mfma_busy {summary['mfma_busy']:.3f}")
    print(f"{prefix}*: {n} launches, {tot / max(1, n) / 1e6:.1f} MB HBM traffic per launch")
    if out_json:
        json.dump(summary, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:])
