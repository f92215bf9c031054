#!/usr/bin/env python3
"""ONE diffusion step of ONE rank's share of the WHOLE-BRAIN sweep, measured on one GPU (BASELINE configs[3] on 8 GPUs:
286 x 414 tiles, row-block partition -> 36 tile rows x 414 columns per rank): the float16 canvas of that share (197 GB), the
16-bit UNet, shared-halo windows of 1 x 4 tiles (35 GiB of workspace: what fits beside the canvas).  The step is taken at
epoch 1 on a random canvas (step 0 would additionally hold a 33 GB band of float32 noise tiles), so neither the halo strips
of the neighbouring ranks nor converged values are involved -- the time of a step does not depend on the values.
  python tools/run_rank_share.py [--rows 36] [--cols 414] [--dtype bf16] [--deadline_s 1100]"""
import argparse
import json
import os
import sys
import time

_T0 = time.monotonic()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=36)
    ap.add_argument("--cols", type=int, default=414)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--batch_tiles", type=int, default=4)
    ap.add_argument("--batch_rows", type=int, default=1, help="tile rows per shared-halo window")
    ap.add_argument("--overlap_streams", type=int, default=1, help="2: two halves of a model call's images on two HIP streams")
    ap.add_argument("--z_group", type=int, default=0, help="images (z-chunks) per model call (0: all 25 of a window at once)")
    ap.add_argument("--deadline_s", type=float, default=1100.0)
    a = ap.parse_args()
    import torch
    import teramind_amd  # noqa: F401
    from teramind_amd import tiles
    from teramind_amd.brain import TileSweep, consistent_gene_provider
    from teramind_amd.config import PathConfig
    from teramind_amd.diffusion import SpacedDiffusionBeatGans
    from teramind_amd.unet import BeatGANsUNetModel
    from teramind_amd.weights import hashed_state_dict
    dev = "cuda:0"
    cfg = PathConfig(gen_type="ddim", compute_dtype=a.dtype)
    model = BeatGANsUNetModel(cfg, dev).load_state_dict(hashed_state_dict(cfg, 0))
    model.overlap_streams = a.overlap_streams
    T = 15
    genes = consistent_gene_provider(cfg, dev, max_blocks=(a.batch_rows + 2) * (a.batch_tiles + 2) + 6,
                                     max_tiles=a.batch_rows * (a.batch_tiles + 2))
    sw = TileSweep(cfg, SpacedDiffusionBeatGans(T, "ddim"), model, genes, hnm=a.rows, wnm=a.cols, total_epochs=T, device=dev,
                   batch_tiles=a.batch_tiles, init="device", state="fp16", share_halo=True, batch_rows=a.batch_rows,
                   z_group=a.z_group or None)
    # a canvas of plausible values in place of the step-0 noise band: rows of N(0, 0.5) clamped to [-1, 1]
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    # (one random 64-row band, repeated down the canvas: the time of a step does not depend on the values, and drawing 98 G
    # normals took minutes of the call's 20)
    band = None
    for y in range(0, sw.cur.shape[1], 64):
        blk = sw.cur[:, y:y + 64]
        if band is None or band.shape != blk.shape:
            band = (torch.randn(blk.shape, generator=g, device=dev, dtype=torch.float32) * 0.5).clamp_(-1, 1).to(sw.cur.dtype)
        blk.copy_(band)
    del band
    sw.epoch = 1
    torch.cuda.synchronize()
    print(f"[rank_share] canvas {tuple(sw.cur.shape)} fp16 = {sw.cur.numel() * 2 / 1e9:.1f} GB, set-up {time.monotonic() - _T0:.0f} s, "
          f"allocated {torch.cuda.memory_allocated() / 2**30:.1f} GiB", file=sys.stderr, flush=True)
    # the step, window by window (TileSweep.step), with a progress line per tile row and a wall-clock guard
    batches = [[(lr, c) for lr in range(l0, min(l0 + sw.batch_rows, sw.nrows)) for c in range(c0, min(c0 + sw.batch_tiles, sw.wnm))]
               for l0 in range(0, sw.nrows, sw.batch_rows) for c0 in range(0, sw.wnm, sw.batch_tiles)]
    t0 = time.perf_counter()
    done_tiles, rows_done = 0, 0
    for batch in batches:
        sw.run_batch(batch, sw.epoch)
        sw._commit_rows(batch[-1][0] - 1 if batch[-1][1] == sw.wnm - 1 else batch[0][0] - 2)
        done_tiles += len(batch)
        if batch[-1][1] == sw.wnm - 1:
            torch.cuda.synchronize()
            rows_done = batch[-1][0] + 1
            el = time.perf_counter() - t0
            print(f"[rank_share] row {rows_done}/{sw.nrows}: {el:.1f} s, {el / done_tiles * 1e3:.2f} ms per tile, peak "
                  f"{torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", file=sys.stderr, flush=True)
            if time.monotonic() - _T0 + 1.1 * el / rows_done > a.deadline_s and rows_done < sw.nrows:
                break
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    full = rows_done == sw.nrows
    st = sw.cur[:, 32:32 + 256, 32:32 + 1024].float()
    per_tile = el / done_tiles
    out = {"what": "one diffusion step of one rank's share of the whole-brain sweep (8-GPU row-block partition), measured on one MI355X",
           "rows": a.rows, "cols": a.cols, "tiles_in_share": a.rows * a.cols, "tiles_measured": done_tiles, "full_step": full,
           "dtype": a.dtype, "state": "fp16 single canvas", "canvas_gb": round(sw.cur.numel() * 2 / 1e9, 1),
           "window_tiles": [a.batch_rows, a.batch_tiles], "z_group": a.z_group or None, "overlap_streams": a.overlap_streams, "share_halo": True,
           "seconds": round(el, 1), "s_per_tile_step": round(per_tile, 5), "interior_patch_steps_per_s": round(400 / per_tile, 1),
           "step_s_for_the_share": round(per_tile * a.rows * a.cols, 1),
           "whole_brain_T15_8gpu_hours_from_this": round(per_tile * a.rows * a.cols * 15 / 3600, 2),
           "peak_allocated_gib": round(torch.cuda.max_memory_allocated() / 2**30, 1),
           "state_finite": bool(torch.isfinite(st).all()), "state_absmax": float(st.abs().max()),
           "note": "the strip exchange with the two neighbouring ranks (2 x 100 x 32 x 106048 fp16 = 1.36 GB per step over xGMI) is not part "
                   "of this single-GPU measurement; rows x cols = ceil(286 / 8) x 414 (tiles.row_block_partition gives the first ranks 36 rows)"}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
