#!/usr/bin/env python3
"""Diagnostic: where does a conv27_bf16 workgroup spend its time?  Runs one test_brn-tile step (bf16) on the diagnostic
library (csrc `make diag`: conv27 with s_memtime stamps at kernel entry / main-loop entry / main-loop exit / kernel exit,
wave 0 of every workgroup) and prints, per launch: workgroups, K stages, median cycles of prologue, main loop (and per
stage), epilogue, and the share of the workgroup lifetime each takes.  The stamps go to a buffer of their own; the
product library contains no stamp code."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["TM_LIB_PATH"] = os.path.join(ROOT, "tera-mind_amd", "csrc", "libteramind_hip_diag%s.so" % os.environ.get("TM_DIAG_ABL", ""))


def main():
    import numpy as np
    import torch
    import teramind_amd  # noqa: F401
    from teramind_amd import _lib, synth
    from teramind_amd.config import PathConfig
    from teramind_amd.unet import BeatGANsUNetModel
    from teramind_amd.weights import hashed_state_dict
    dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    b, P = (25, 4) if (len(sys.argv) < 3 or sys.argv[2] == "tile") else (32, 1)
    dev = torch.device("cuda:0")
    cfg = PathConfig(compute_dtype=dtype)
    m = BeatGANsUNetModel(cfg, dev).load_state_dict(hashed_state_dict(cfg, 0))
    L = _lib.lib()
    L.tm_diag_stamps.restype, L.tm_diag_stamps.argtypes = C.c_int, [C.c_void_p, C.c_uint]
    L.tm_diag_stamp_count.restype, L.tm_diag_stamp_count.argtypes = C.c_int, []
    p = P + 1
    x = synth.normal("st/x", (b * p * p, 4, 64, 64), 5).to(dev)
    rna = synth.gene_counts("st/rna", (b * p * p, 4, 4, 2000), 5).to(dev)
    t = torch.full((b,), 601, dtype=torch.long, device=dev)
    kw = dict(x=x, t=t, rna=rna, imgs=torch.empty((b, 4, 64 * P, 64 * P), device="meta"), patch_size=64)
    for _ in range(2):
        m(**kw)
    torch.cuda.synchronize()
    cap = 800_000
    buf = torch.zeros((cap, 16), dtype=torch.int64, device=dev)
    assert L.tm_diag_stamps(C.c_void_p(buf.data_ptr()), cap) == 0
    m(**kw)
    torch.cuda.synchronize()
    n = L.tm_diag_stamp_count()
    L.tm_diag_stamps(None, 0)
    a = buf[:min(n, cap)].cpu().numpy()
    # the ping-pong kernel stamps lane 0 of wave 0 AND of wave 4 (tag + 4): keep wave 0's slots for this table
    a = a[((a[:, 6] % 10) // 4) == 0]
    # split into launches: consecutive slots with the same (grid, tag) up to `grid` entries
    out, i = [], 0
    while i < len(a):
        grid, tag = int(a[i, 4]), int(a[i, 6])
        j = i
        while j < len(a) and j - i < grid and int(a[j, 4]) == grid and int(a[j, 6]) == tag:
            j += 1
        s = a[i:j]
        pro, loop, epi = s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2]
        life = s[:, 3] - s[:, 0]
        cbp, rest = divmod(tag, 1000000)
        is1 = rest >= 500000                                           # conv1 (1x1x1 / Linear) launch
        tn, rest = divmod(rest - (500000 if is1 else 0), 1000)
        tw, fuse = divmod(rest, 10)
        nh = (cbp + (tn // 64) - 1) // (tn // 64) if is1 else 2 * cbp   # conv1: stages of KP = TN/64 pairs
        span_us = (s[:, 7].max() - s[:, 7].min()) / 100.0              # s_memrealtime: 100 MHz
        out.append({"kind": "conv1" if is1 else "conv27", "TN": tn, "TW": tw, "fuse": fuse, "pairs": cbp, "stages": nh, "wgs": int(j - i), "grid": grid,
                    "prologue": float(np.median(pro)), "loop": float(np.median(loop)), "loop_per_stage": float(np.median(loop)) / nh,
                    "epilogue": float(np.median(epi)), "life": float(np.median(life)),
                    "epilogue_p90": float(np.percentile(epi, 90)), "prologue_p90": float(np.percentile(pro, 90)),
                    "first_to_last_start_us": float(span_us)})
        i = j
    print(f"{'TN':>4s} {'TW':>3s} {'f':>1s} {'pairs':>5s} {'wgs':>6s} | {'prologue':>9s} {'loop':>9s} {'/stage':>7s} {'epilogue':>9s} {'life':>9s} | "
          f"{'pro%':>5s} {'loop%':>6s} {'epi%':>5s} | ideal/stage 4608 (8 waves) or 2304x2 (4 waves)")
    tot = {"pro": 0.0, "loop": 0.0, "epi": 0.0, "ideal": 0.0}
    for r in out:
        print(("c1 " if r["kind"] == "conv1" else "c27") + f"{r['TN']:4d} {r['TW']:3d} {r['fuse']:1d} {r['pairs']:5d} {r['wgs']:6d} | {r['prologue']:9.0f} {r['loop']:9.0f} {r['loop_per_stage']:7.0f} "
              f"{r['epilogue']:9.0f} {r['life']:9.0f} | {100 * r['prologue'] / r['life']:5.1f} {100 * r['loop'] / r['life']:6.1f} "
              f"{100 * r['epilogue'] / r['life']:5.1f}")
        if r["kind"] == "conv27":
            tot["pro"] += r["prologue"] * r["wgs"]; tot["loop"] += r["loop"] * r["wgs"]; tot["epi"] += r["epilogue"] * r["wgs"]
        if r["kind"] == "conv1":
            continue
        tot["ideal"] += 4608.0 * r["stages"] * r["wgs"]
    s_ = tot["pro"] + tot["loop"] + tot["epi"]
    print(f"workgroup-cycles: prologue {100 * tot['pro'] / s_:.1f} %  main loop {100 * tot['loop'] / s_:.1f} %  epilogue {100 * tot['epi'] / s_:.1f} %;"
          f"  MFMA-ideal share of the whole {100 * tot['ideal'] / s_:.1f} %  (of the main loop {100 * tot['ideal'] / tot['loop']:.1f} %)")
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "conv27_stamps.json"), "w"))


if __name__ == "__main__":
    main()
