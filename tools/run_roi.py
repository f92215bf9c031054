#!/usr/bin/env python3
"""End-to-end rehearsal of the test_brn flow on one GPU with synthetic genes and hashed weights: a full T-step
DDIM sweep of an hnm x wnm tile ROI through the product classes (TileSweep + SpacedDiffusionBeatGans +
BeatGANsUNetModel), then save_step -> stitch_dir -> slice images, exactly the sequence
`python -m test_brn ...` + `python -m infer_brn ...` performs in the reference.  Prints one JSON line with the
measured wall time (this is a measurement at the stated ROI size, not an extrapolation)."""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hnm", type=int, default=4)
    ap.add_argument("--wnm", type=int, default=8)
    ap.add_argument("--tot_epoch", type=int, default=15)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="bf16")
    ap.add_argument("--state", choices=["fp32x2", "fp16"], default="fp16")
    ap.add_argument("--out_dir", default=None)
    args = ap.parse_args()
    import numpy as np
    import torch
    import teramind_amd  # noqa: F401
    from teramind_amd import stitch
    from teramind_amd.brain import TileSweep, synthetic_gene_provider
    from teramind_amd.config import PathConfig
    from teramind_amd.diffusion import SpacedDiffusionBeatGans
    from teramind_amd.unet import BeatGANsUNetModel
    from teramind_amd.weights import hashed_state_dict

    dev = "cuda:0"
    cfg = PathConfig(compute_dtype=args.dtype)
    model = BeatGANsUNetModel(cfg, dev).load_state_dict(hashed_state_dict(cfg, 0))
    host_genes, resident = synthetic_gene_provider(cfg, total_slc=50), {}

    def genes(row, col):
        if (row, col) not in resident:
            resident[(row, col)] = host_genes(row, col).to(dev)
        return resident[(row, col)]

    T = args.tot_epoch
    sw = TileSweep(cfg, SpacedDiffusionBeatGans(T, "ddim"), model, genes, hst=256, wst=256, hnm=args.hnm, wnm=args.wnm,
                   total_epochs=T, total_slc=50, device=dev, batch_tiles=1, init="device", state=args.state)
    for r in range(args.hnm):
        for c in range(args.wnm):
            genes(1 + r, 1 + c)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    per_step = []
    while sw.epoch < T:
        t1 = time.perf_counter()
        sw.step()
        torch.cuda.synchronize()
        per_step.append(time.perf_counter() - t1)
        print(f"[run_roi] step {sw.epoch}/{T}: {per_step[-1]:.2f} s", file=sys.stderr, flush=True)
    sweep_s = time.perf_counter() - t0
    st = sw.local_state()
    out_dir = args.out_dir or tempfile.mkdtemp(prefix="roi_")
    t2 = time.perf_counter()
    d = sw.save_step(os.path.join(out_dir, "timestep"))
    mosaic = stitch.stitch_dir(d, 256, 256, args.hnm, args.wnm, 50, slices=[0, 1, 48, 49])
    stitch.save_slices(mosaic, os.path.join(out_dir, "gen"), names=[0, 1, 48, 49])
    io_s = time.perf_counter() - t2
    same = bool(np.array_equal(mosaic, stitch.stitch_state(st, 50, [0, 1, 48, 49]).cpu().numpy()))
    tiles = args.hnm * args.wnm
    print(json.dumps({"what": "full ROI sweep, measured", "dtype": args.dtype, "state": args.state, "tiles": tiles, "T": T,
                      "sweep_s": round(sweep_s, 2), "s_per_tile_step": round(sweep_s / (tiles * T), 4),
                      "interior_patch_steps_per_s": round(400 * tiles * T / sweep_s, 1),
                      "first_step_s": round(per_step[0], 2), "last_step_s": round(per_step[-1], 2),
                      "save_and_stitch_s": round(io_s, 2), "stitch_from_files_equals_resident": same,
                      "state_finite": bool(torch.isfinite(st.float()).all()), "state_absmax": float(st.float().abs().max()),
                      "state_std": float(st.float().std())}))


if __name__ == "__main__":
    main()
