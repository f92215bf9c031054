#!/usr/bin/env python3
"""One test_brn tile forward (b = 25 z-chunks, P = 4) on one stream vs `model.overlap_streams = 2` (13 + 12 z-chunks on two HIP
streams with a workspace each), with dense genes and with the cached level 0 of the RNA conditioning.  Kernels of the two streams may overlap: one stream's partial last
round of workgroups (and its HBM-bound block-input passes / Linears) beside the other's MFMA-bound convs.
   python3 tools/two_stream_check.py [bf16|f16|f32]
Prints ms per tile step for both forms (interleaved rounds, same process) and checks that the outputs are bit-identical."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import teramind_amd  # noqa: F401
    from teramind_amd import synth
    from teramind_amd.config import PathConfig
    from teramind_amd.unet import BeatGANsUNetModel
    from teramind_amd.weights import hashed_state_dict
    dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    dev = torch.device("cuda:0")
    cfg = PathConfig(compute_dtype=dtype)
    m = BeatGANsUNetModel(cfg, dev).load_state_dict(hashed_state_dict(cfg, 0))
    b, P = 25, 4
    p = P + 1
    x = synth.normal("st/x", (b * p * p, 4, 64, 64), 5).to(dev)
    rna = synth.gene_counts("st/rna", (b * p * p, 4, 4, 2000), 5).to(dev)
    t = torch.full((b,), 601, dtype=torch.long, device=dev)
    imgs = torch.empty((b, 4, 64 * P, 64 * P), device="meta")
    l0 = m.precompute_rna_level0(rna, b, imgs=imgs, patch_size=64)
    torch.cuda.synchronize()
    variants = {"full": (1, rna), "split": (2, rna), "full_l0": (1, l0), "split_l0": (2, l0)}

    def call(name):
        m.overlap_streams, r = variants[name]
        return m(x=x, t=t, rna=r, imgs=imgs, patch_size=64).pred

    # warm up (allocates the workspaces)
    ref = call("full")
    for name in ("split", "full_l0", "split_l0"):
        got = call(name)
        torch.cuda.synchronize()
        print(f"{name:9s} bit-identical to the one-stream dense-gene call:", bool(torch.equal(ref, got)))
    res = {k: [] for k in variants}
    for rnd in range(4):
        for name in variants:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                call(name)
            torch.cuda.synchronize()
            res[name].append((time.perf_counter() - t0) / 3 * 1e3)
    for k, v in res.items():
        print(f"{k:9s} ms per tile forward: " + "  ".join(f"{a:7.2f}" for a in v) + f"   median {sorted(v)[len(v) // 2]:.2f}")


if __name__ == "__main__":
    main()
