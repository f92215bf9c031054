#!/usr/bin/env python3
"""End-to-end run of the test_brn flow with synthetic genes and hashed weights: a full T-step DDIM sweep of an
hnm x wnm tile ROI through the product classes (launch.run_sweep -> TileSweep + SpacedDiffusionBeatGans +
BeatGANsUNetModel), then save_step -> stitch_dir -> slice images, exactly the sequence `python -m test_brn ...` +
`python -m infer_brn ...` performs in the reference (test_brn.py:232-273, infer_brn.py:57-105).
`--gpus N` starts N rank processes (one per GPU; rows of the tile grid are split over them, RCCL strip exchange per
step).  `--gene_dir` reads the reference's on-disk gene tiles ('{r0}_..._{C1}.npz' COO archives, made by
`--make_genes` from the synthetic generator) through brain.GeneTileDir: the COO arrays stay resident and
tm_gene_tile_dense re-densifies them for every tile call.  Prints one JSON line with the measured wall time (a
measurement at the stated ROI size, not an extrapolation)."""
import argparse
import json
import os
import sys
import tempfile
import time

_T_START = time.monotonic()

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_gene_dir(gdir, hnm, wnm, nnz, hst=256, wst=256):
    """Synthetic COO gene tiles in the reference's on-disk format, cut from ONE gene map (synth.write_gene_tile_dir): tiles that
    overlap agree, so --share_halo is legitimate on them (TileSweep checks it)."""
    from teramind_amd import synth
    synth.write_gene_tile_dir(gdir, hnm, wnm, max(1, nnz // 16), hst=hst, wst=wst)


def worker(args):
    import numpy as np
    import torch
    import teramind_amd  # noqa: F401
    from teramind_amd import launch, stitch
    from teramind_amd.brain import GeneTileDir, consistent_gene_provider, device_gene_provider
    from teramind_amd.config import PathConfig
    from teramind_amd.diffusion import SpacedDiffusionBeatGans
    from teramind_amd.unet import BeatGANsUNetModel
    from teramind_amd.weights import hashed_state_dict

    rank, local_rank, world = launch.dist_env()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    launch.init_distributed("nccl", dev)
    cfg = PathConfig(compute_dtype=args.dtype)
    model = BeatGANsUNetModel(cfg, dev).load_state_dict(hashed_state_dict(cfg, 0))
    model.overlap_streams = args.overlap_streams
    T = args.tot_epoch
    holder = {}
    if args.deadline_s and world > 1:
        raise SystemExit("--deadline_s is for single-GPU runs")
    if args.gene_dir:
        genes = GeneTileDir(args.gene_dir, cfg, dev, total_slc=50, keep_resident=True)
    else:
        genes = consistent_gene_provider(cfg, dev) if args.share_halo else device_gene_provider(cfg, dev)

    class Deadline(Exception):
        pass

    def on_step(sw, s):
        if rank == 0:
            print(f"[run_roi] step {sw.epoch}/{T}: {s:.2f} s  (process age {time.monotonic() - _T_START:.0f} s)", file=sys.stderr, flush=True)
        # the GPU runner ends a command at a fixed wall-clock limit: never start a step that cannot finish inside it
        if args.deadline_s and sw.epoch < T and time.monotonic() - _T_START + 1.03 * s > args.deadline_s:
            raise Deadline()

    steps_done = T
    try:
        res = launch.run_sweep(cfg, SpacedDiffusionBeatGans(T, "ddim"), model, genes, hnm=args.hnm, wnm=args.wnm, total_epochs=T,
                               steps=T, warmup=0, device=dev, batch_tiles=args.batch_tiles, init=args.init, state=args.state,
                               on_step=on_step, holder=holder, share_halo=bool(args.share_halo),
                               batch_rows=args.batch_rows, prefetch_genes=not args.share_halo,
                               cache_level0=bool(args.cache_level0))
    except Deadline:
        # single-rank runs only (a rank that stops alone would leave its neighbours in the strip exchange)
        sw = holder["sweep"]
        torch.cuda.synchronize()
        steps_done = sw.epoch
        res = {"sweep": sw, "dt": sum(holder["step_s"]), "step_s": holder["step_s"], "exchange_ms_per_step": 0.0}
    sw, sweep_s, per_step = res["sweep"], res["dt"], res["step_s"]
    st = sw.local_state()
    out_dir = args.out_dir or tempfile.mkdtemp(prefix="roi_")
    t2 = time.perf_counter()
    io_save, d = 0.0, None
    if not args.no_tile_files:
        d = sw.save_step(os.path.join(out_dir, "timestep"))
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        io_save = time.perf_counter() - t2
    if rank == 0:
        t3 = time.perf_counter()
        same = None
        if d is not None:
            mosaic = stitch.stitch_dir(d, 256, 256, args.hnm, args.wnm, 50, slices=[0, 1, 48, 49])
            if world == 1:
                same = bool(np.array_equal(mosaic, stitch.stitch_state(st, 50, [0, 1, 48, 49]).cpu().numpy()))
        else:                                   # mosaic straight from the resident canvas (infer_brn.py:57-105 on device)
            mosaic = stitch.stitch_state(st, 50, [0, 1, 48, 49]).cpu().numpy()
        stitch.save_slices(mosaic, os.path.join(out_dir, "gen"), names=[0, 1, 48, 49])
        io_s = io_save + time.perf_counter() - t3
        tiles = args.hnm * args.wnm
        T = steps_done
        print(json.dumps({"what": "full ROI sweep, measured", "steps_requested": args.tot_epoch, "steps_measured": steps_done,
                          "step_s": [round(v, 2) for v in per_step], "process_age_s": round(time.monotonic() - _T_START, 1),
                          "tile_files_written": d is not None, "dtype": args.dtype, "state": args.state, "tiles": tiles, "T": T,
                          "n_gpus": world, "genes": "on-disk COO .npz via GeneTileDir + tm_gene_tile_dense" if args.gene_dir else "synthetic, device resident",
                          "init": args.init, "batch_tiles": args.batch_tiles, "share_halo": bool(args.share_halo), "batch_rows": args.batch_rows, "cache_level0": bool(args.cache_level0), "overlap_streams": args.overlap_streams,
                          "sweep_s": round(sweep_s, 2), "sweep_min": round(sweep_s / 60, 2), "s_per_tile_step": round(sweep_s * world / (tiles * T), 4),
                          "interior_patch_steps_per_s": round(400 * tiles * T / sweep_s, 1),
                          "first_step_s": round(per_step[0], 2), "last_step_s": round(per_step[-1], 2),
                          "exchange_ms_per_step": round(res["exchange_ms_per_step"], 3),
                          "save_and_stitch_s": round(io_s, 2), "stitch_from_files_equals_resident": same,
                          "state_finite": bool(torch.isfinite(st.float()).all()), "state_absmax": float(st.float().abs().max()),
                          "state_std": float(st.float().std())}), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--hnm", type=int, default=4)
    ap.add_argument("--wnm", type=int, default=8)
    ap.add_argument("--tot_epoch", type=int, default=15)
    ap.add_argument("--dtype", choices=["f32", "bf16", "f16"], default="bf16")
    ap.add_argument("--state", choices=["fp32x2", "fp16"], default="fp16")
    ap.add_argument("--init", choices=["device", "reference"], default="device",
                    help="reference = the LCG-seeded CPU mt19937 noise of MBADataset_tst (slow: ~0.1 s per tile)")
    ap.add_argument("--batch_tiles", type=int, default=1)
    ap.add_argument("--batch_rows", type=int, default=1, help="tile rows per model call (with --share_halo 1)")
    ap.add_argument("--overlap_streams", type=int, default=1, help="2: two halves of a call's images on two HIP streams (bit-identical)")
    ap.add_argument("--cache_level0", type=int, default=0,
                    help="1: keep level 0 of the RNA conditioning of every model call across the steps (ROI scale: 27-37 MB per tile)")
    ap.add_argument("--share_halo", type=int, default=0,
                    help="1: the tiles of a model call form ONE window, patch columns shared by neighbouring tiles go through the "
                         "encoder once (TileSweep(share_halo=True); needs gene tiles that agree where they overlap)")
    ap.add_argument("--gene_dir", default=None)
    ap.add_argument("--make_genes", type=int, default=0, help="write synthetic COO gene tiles with this many entries each into --gene_dir first")
    ap.add_argument("--out_dir", default=None)
    ap.add_argument("--deadline_s", type=float, default=0.0,
                    help="single GPU: do not start a diffusion step that would end later than this many seconds after process start")
    ap.add_argument("--no_tile_files", action="store_true", help="skip save_step / stitch_dir; stitch slice images from the resident canvas")
    args = ap.parse_args()
    from teramind_amd import launch
    if args.gene_dir and args.make_genes and not launch.launched_as_rank():
        t0 = time.perf_counter()
        make_gene_dir(args.gene_dir, args.hnm, args.wnm, args.make_genes)
        print(f"[run_roi] wrote {args.hnm * args.wnm} gene tiles in {time.perf_counter() - t0:.1f} s", file=sys.stderr, flush=True)
    if args.gpus > 1 and not launch.launched_as_rank():
        sys.exit(launch.spawn_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))
    worker(args)


if __name__ == "__main__":
    main()
