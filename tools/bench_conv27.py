"""Times the 16-bit 3x3x3 conv (tm_op_conv27_time) on the ResBlock conv shapes of one `test_brn` tile step
(b = 25 z-chunks, P = 4: 625 encoder / 400 decoder patches) for several kernel forms IN ONE PROCESS, interleaved rounds
(cdna guide rule 24), random operands.  `--forms 8,9` = ping-pong vs lockstep 8-wave kernel.

  python tools/bench_conv27.py --forms 8,9 --rounds 3 --iters 5 [--dtype bf16] [--layers small] [--json out.json]

Executed FLOPs = 2/3 of 2*Cin*Cout*27*voxels (Z = 2: the third z tap only meets padding and is never issued); for the
upsampled-input form 8 taps of the low-resolution tensor per output phase."""
import argparse
import ctypes as C
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import teramind_amd  # noqa: F401,E402
from teramind_amd import _lib  # noqa: E402

# (name, Cin, Cout, S, N, res, fused, ups): the 53 counted launches of a tile step collapse to these distinct shapes
# (count = how many launches of a step have the shape)
LAYERS = [
    ("enc0 c1 96>64 S64", 96, 64, 64, 625, 0, 1, 0, 2),
    ("enc0 c2 64>64 S64", 64, 64, 64, 625, 1, 0, 0, 2),
    ("down0 64>64 S32", 64, 64, 32, 625, 0, 1, 0, 2),
    ("enc1 c1 192>128 S32", 192, 128, 32, 625, 0, 1, 0, 1),
    ("enc1 c2 128>128 S32", 128, 128, 32, 625, 1, 0, 0, 3),
    ("down1 128>128 S16", 128, 128, 16, 625, 0, 1, 0, 2),
    ("enc2 c1 384>256 S16", 384, 256, 16, 625, 0, 0, 0, 1),
    ("enc2 c2 256>256 S16", 256, 256, 16, 625, 1, 0, 0, 3),
    ("down2 256>256 S8", 256, 256, 8, 625, 0, 0, 0, 2),
    ("enc3 c1 741>512 S8", 741, 512, 8, 625, 0, 0, 0, 2),
    ("enc3 c2 512>512 S8", 512, 512, 8, 625, 1, 0, 0, 6),
    ("dec3 c1 1253>512 S8", 1253, 512, 8, 400, 0, 0, 0, 2),
    ("dec3 c2 512>512 S8", 512, 512, 8, 400, 1, 0, 0, 3),
    ("up3 c2 512>512 S16", 512, 512, 16, 400, 1, 0, 0, 1),
    ("up3 c1 512>512 S8 ups", 512, 512, 8, 400, 0, 0, 1, 1),
    ("up2 c1 256>256 S16 ups", 256, 256, 16, 400, 0, 0, 1, 1),
    ("up1 c1 128>128 S32 ups", 128, 128, 32, 400, 0, 1, 1, 1),
    ("dec2 c1 896>256 S16", 896, 256, 16, 400, 0, 0, 0, 1),
    ("dec2 c2 256>256 S16", 256, 256, 16, 400, 1, 0, 0, 3),
    ("up2 c2 256>256 S32", 256, 256, 32, 400, 1, 0, 0, 1),
    ("dec1 c1 448>128 S32", 448, 128, 32, 400, 0, 1, 0, 1),
    ("dec1 c2 128>128 S32", 128, 128, 32, 400, 1, 0, 0, 3),
    ("up1 c2 128>128 S64", 128, 128, 64, 400, 1, 0, 0, 1),
    ("dec0 c1 224>64 S64", 224, 64, 64, 400, 0, 1, 0, 1),
    ("dec0 c1 160>64 S64", 160, 64, 64, 400, 0, 1, 0, 2),
    ("dec0 c2 64>64 S64", 64, 64, 64, 400, 1, 0, 0, 3),
]
SMALL = {"enc1 c2 128>128 S32", "enc2 c2 256>256 S16", "enc3 c2 512>512 S8", "dec3 c1 1253>512 S8", "dec0 c1 160>64 S64",
         "dec0 c2 64>64 S64", "up3 c1 512>512 S8 ups", "up2 c2 256>256 S32"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--forms", default="8,9")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--layers", default="all")
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    forms = [int(f) for f in args.forms.split(",")]
    dt = {"bf16": 1, "f16": 2}[args.dtype]
    L = _lib.lib()
    torch.zeros(1, device="cuda")
    st = _lib.current_stream_ptr()
    ms = C.c_float(0)
    rows = []
    tot = {f: 0.0 for f in forms}
    totflop = 0.0
    for (name, cin, cout, S, N, res, fused, ups, cnt) in LAYERS:
        if args.layers == "small" and name not in SMALL:
            continue
        t = {f: [] for f in forms}
        for _ in range(args.rounds):
            for f in forms:
                _lib.check(L.tm_op_conv27_time(N, cin, cout, S, dt, f, ups, res, fused, args.iters, C.byref(ms), st), "tm_op_conv27_time")
                t[f].append(ms.value)
        taps = 8 if ups else 18
        so = 2 * S if ups else S
        flops = 2.0 * cin * cout * taps * N * 2 * so * so
        med = {f: statistics.median(t[f]) for f in forms}
        rows.append({"layer": name, "count": cnt, "exec_gflop": flops / 1e9,
                     "ms": med, "tflops": {f: flops / med[f] / 1e9 for f in forms}, "min_ms": {f: min(t[f]) for f in forms}})
        for f in forms:
            tot[f] += cnt * med[f]
        totflop += cnt * flops
        print(f"{name:26s} x{cnt} " + "  ".join(f"[{f}] {med[f]:7.3f} ms {flops / med[f] / 1e9:7.1f} TF" for f in forms), flush=True)
    print("weighted sum: " + "  ".join(f"[{f}] {tot[f]:8.3f} ms {totflop / tot[f] / 1e9:7.1f} TF ({totflop / tot[f] / 1e9 / 2500:.3f} of 2500)" for f in forms))
    if args.json:
        with open(args.json, "w") as fh:
            json.dump({"dtype": args.dtype, "forms": forms, "rounds": args.rounds, "iters": args.iters, "rows": rows,
                       "weighted_ms": tot, "weighted_tflops": {f: totflop / tot[f] / 1e9 for f in forms}}, fh, indent=1)


if __name__ == "__main__":
    main()
