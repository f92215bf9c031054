#!/usr/bin/env python3
"""Times the block-input pass of the 16-bit modes (tm_op_prep_h16) on the shapes of one test_brn tile step, in each kernel
form (1 = prep_kernel, 2 / 3 = prep_h16_kernel with one wave / four waves per 64 voxels), and prints the HBM rate from the
algorithmic bytes (every source element read once, every output element written once, modulation tensors read once).
  python tools/bench_prep.py [--dtype bf16] [--iters 20]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch          # noqa: E402
import util           # noqa: E402

# (label, b, p1, p2, cins, collage flags, S, up2, mod, norm): test_brn tile, 25 z-chunks, 5 x 5 padded patches per chunk
CASES = [
    ("enc L0 cat(h64, rna32)", 25, 5, 5, (64, 32), (0, 0), 64, False, 0, True),
    ("enc L1 cat(h64, rna64)", 25, 5, 5, (64, 64), (0, 0), 32, False, 0, True),
    ("enc L1 cat(h128, rna64)", 25, 5, 5, (128, 64), (0, 0), 32, False, 0, True),
    ("attn L1 modulate 128", 25, 5, 5, (128,), (0,), 32, False, 2, True),
    ("enc L2 cat(h256, rna128)", 25, 5, 5, (256, 128), (0, 0), 16, False, 0, True),
    ("enc L3 cat(h512, rna229)", 25, 5, 5, (512, 229), (0, 0), 8, False, 0, True),
    ("out_layers 512 (per-image mod)", 25, 5, 5, (512,), (0,), 8, False, 1, True),
    ("dec L3 cat(512, 512, 229) collage", 25, 5, 5, (512, 512, 229), (0, 1, 1), 8, False, 0, True),
    ("dec L2 cat(512, 256, 128) collage", 25, 5, 5, (512, 256, 128), (0, 1, 1), 16, False, 0, True),
    ("dec L1 cat(256, 128, 64) collage", 25, 5, 5, (256, 128, 64), (0, 1, 1), 32, False, 0, True),
    ("dec L0 cat(128, 64, 32) collage", 25, 5, 5, (128, 64, 32), (0, 1, 1), 64, False, 0, True),
    ("dec L0 cat(64, 64, 32) collage", 25, 5, 5, (64, 64, 32), (0, 1, 1), 64, False, 0, True),
    ("up L0 <- L1 (128)", 25, 5, 5, (128,), (0,), 64, True, 0, True),
    ("SiLU(cond 64) collage", 25, 5, 5, (64,), (1,), 32, False, 0, False),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--forms", default="1,2,3,0")
    a = ap.parse_args()
    dev = "cuda:0"
    td = util.H16[a.dtype][1]
    forms = [int(v) for v in a.forms.split(",")]
    print(f"{'case':38s} {'GB':>6s} " + " ".join(f"{'form ' + str(v) + ' us':>11s} {'TB/s':>5s}" for v in forms))
    for label, b, p1, p2, cins, flags, S, up2, mod, norm in CASES:
        g = torch.Generator(device=dev).manual_seed(1)
        any_col = any(flags)
        # decoder blocks: the plain first source already lives on the (p1-1) x (p2-1) collage grid
        Ne, Nd = b * p1 * p2, b * (p1 - 1) * (p2 - 1)
        N = Nd if any_col else Ne
        Ss = S // 2 if up2 else S
        xs = [torch.randn((Ne if f else N, c, 2, Ss, Ss), generator=g, device=dev).to(td).float() for c, f in zip(cins, flags)]
        ws = [torch.ones((c,)) for c in cins] if norm else None
        Ct = sum(cins)
        scale = shift = None
        if mod == 1:
            scale, shift = torch.zeros((b, Ct)), torch.zeros((b, Ct))
        elif mod == 2:
            scale = torch.zeros((N, Ct, 2, S, S), device=dev)
            shift = torch.zeros((N, Ct, 2, S, S), device=dev)
        vox = N * 2 * S * S
        cb = sum((c + 7) // 8 for c in cins)
        nbytes = vox * cb * 8 * 2 * (2 + (2 if mod == 2 else 0)) if not up2 else vox * cb * 8 * 2 * 1.25
        cols = []
        for v in forms:
            if v == 2 and cb > 32:
                cols.append(f"{'-':>11s} {'-':>5s}")
                continue
            _, _, ms = util.prep_h16(xs, cins, flags, b, p1, p2, S, up2=up2, norm_w=ws, mod=mod, scale=scale, shift=shift,
                                     per_image=N // b, act=mod != 2, dtype=a.dtype, variant=v, iters=a.iters)
            cols.append(f"{ms * 1e3:11.1f} {nbytes / ms / 1e9:5.2f}")
        print(f"{label:38s} {nbytes / 1e9:6.2f} " + " ".join(cols), flush=True)
        del xs, scale, shift
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
