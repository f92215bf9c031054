#!/usr/bin/env python3
"""Headline benchmark of the denoising hot path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W            (starts its own N worker processes when N > 1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Default workload (the headline `value`): a "step" is one denoising step of the hot path over one batch of synthetic
input: pad+patchify -> UNet (tm_unet_forward) -> DDPM update (tm_sampler_step) for b=32 images of one 64x64 interior
patch each (BASELINE configs[1]: batch_size=32, patch_size=64, rna_slc=4, T=50 DDPM, fp32; mode A, P=1 => 128 padded
encoder patches, 32 collage decoder patches per step).  All inputs are resident in HBM before the timed region.  The
unit is the interior patch-step.  With N > 1 every rank runs its own batch of 32 patches (the
path shards by independent patches: weak scaling, no data-path collective).

Row-sharded tile sweep (`--sweep`, and a short one appended to every default run as the `sweep` object): a fixed ROI of
`--sweep-hnm x --sweep-wnm` test_brn tiles (256x256 px x 100 channels, 25 z-chunks x 16 interior patches each) in the
whole-brain configuration (16-bit UNet arithmetic, one float16 state canvas per rank), row-sharded over the N ranks by
teramind_amd.launch.run_sweep / brain.TileSweep: rank 0's packed weight arena is broadcast once over RCCL and every
diffusion step ends with the 32-px halo-strip exchange with the neighbouring ranks (send / recv over xGMI).  Strong
scaling: the ROI is the same for every N.  Reported: interior patch-steps/s of the whole job, per-step exchange time
and bytes, and the world size as RCCL sees it.

The default line also carries `sweep.roofline` (the 16-bit conv kernel's executed TFLOP/s, fraction of the dense peak and
counter traffic during the sweep's timed steps) and `extra.tile_<dtype>`: the stacked one-tile step (BASELINE config 4's
governing shape) on the sweep's packed model, 3 timed steps, with its own `roofline`.

Prints ONE JSON line on rank 0, with `roofline` (dominant kernel, the 3x3x3 implicit-GEMM conv: MFMA FLOPs actually
issued per launch / average launch duration from hipEvents recorded on the launch stream during the timed region,
against the dense MFMA peak of the dtype) and `cpu_baseline` (oracle/teramind_cpu.py on the host cores, bounded sample).
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0    # same table, "Peak BF16/FP16 MFMA ~2.5 PF dense"
T_STEPS = 50
NEEDED_GFLOP = {1: 320.1, 4: 202.6}   # needed FLOPs per interior patch-step (whole path, 2*MAC)
Z_AWARE = 0.684                       # share of those FLOPs left when the always-zero z tap is not issued


_T0 = time.perf_counter()


def log(msg):
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def kernel_source_sha() -> str:
    """sha256 over the kernel sources: `roofline.traffic` (PMC counters, collected in separate rocprofv3 passes and kept
    under profiles/) is only reported when it was measured on exactly these sources."""
    d = os.path.join(ROOT, "tera-mind_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")) or f == "Makefile":
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(cfg, sd, budget_steps=6, b=8, P=1):
    """Oracle (CPU restatement) timed on this host: `budget_steps` full steps (UNet + DDPM update)
    of b images x 1 interior patch after one warm-up forward."""
    import torch
    from oracle import teramind_cpu as tc
    from teramind_amd import synth
    oc = tc.oracle_config_from(cfg)
    sch = tc.make_schedule(T_STEPS, "ddpm")
    # the GPU box gives a one-GPU job a 16-core share; os.cpu_count() reports the whole host and
    # oversubscribing oneDNN by 10x stalls for minutes
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, 16))
    torch.set_num_threads(threads)
    log(f"cpu_baseline: {threads} threads (affinity {avail}, cpu_count {os.cpu_count()})")
    ne = b * (P + 1) ** 2
    x = synth.normal("cpu/x", (ne, cfg.in_channels, 64, 64), 0)
    rna = synth.gene_counts("cpu/rna", (ne, cfg.gn_sz, cfg.gn_sz, cfg.rna_slc * 500), 0)
    nz = synth.normal("cpu/nz", (ne, cfg.in_channels, 64, 64), 1)
    t = torch.full((b,), sch.timestep_map[T_STEPS - 1], dtype=torch.long)
    with torch.inference_mode():
        tc.unet_forward(sd, oc, x[:4], t[:1], rna[:4], 2, 2)           # warm-up
        log("cpu_baseline: warm-up forward done")
        t0 = time.perf_counter()
        for k in range(budget_steps):
            i = T_STEPS - 1 - k
            pred, _ = tc.unet_forward(sd, oc, x, t, rna, P + 1, P + 1)
            img = tc.sampler_step(sch, "ddpm", x, pred, i, P, P, nz)
            x = tc.patchify(torch.nn.functional.pad(img, (32, 32, 32, 32)), 64)
        dt = time.perf_counter() - t0
    return {"value": round(b * budget_steps / dt, 4), "unit": "interior patch-steps/s", "cores": threads,
            "kind": "port",
            "sample": f"{budget_steps} DDPM steps of b={b} images x 1 interior 64x64 patch (P=1, {ne} padded patches), "
                      f"fp32 torch CPU oracle, {dt:.1f} s"}


def roofline_block(prof, dtype, dt_total, value_per_gpu, P, tile, sweep=False):
    """`roofline` of the dominant kernel from the hipEvent brackets of the timed region (tm_profile_collect)."""
    peak = PEAK_F32_MFMA_TFLOPS if dtype == "f32" else PEAK_BF16_MFMA_TFLOPS
    ms = prof["total_ms"]
    n = max(1, prof["launches"])
    tfl = lambda fl: fl / (ms * 1e-3) / 1e12 if ms else 0.0
    executed, nominal = tfl(prof["executed_flops"]), tfl(prof["nominal_flops"])
    traffic, traffic_src = None, None
    pmc_json = os.path.join(ROOT, "profiles", "conv27_traffic" + ("_bf16" if dtype != "f32" else "") +
                            ("_tile" if (tile or sweep) else "") + ".json")
    if os.path.exists(pmc_json):
        try:
            rec = json.load(open(pmc_json))
            if rec.get("src_sha") == kernel_source_sha():
                traffic = rec.get("hbm_bytes_per_launch")
                traffic_src = (f"{os.path.relpath(pmc_json, ROOT)} @ {rec.get('commit', '?')} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of " +
                               ("the ONE-TILE step: per launch of one tile's patches -- a launch of this sweep covers a whole window of tiles, "
                                "scale by executed_gflop_per_launch / 851.3)" if sweep else "this workload)"))
            else:
                traffic_src = f"{os.path.relpath(pmc_json, ROOT)} is stale (kernel sources changed since it was measured): not reported"
        except Exception:
            pass
    whole = NEEDED_GFLOP[P] * Z_AWARE * value_per_gpu / 1e3          # z-aware needed TFLOP/s of the whole step, per GPU
    return {"bound": "mfma",
            "kernel": ("conv3d_mfma<2,*,*> (3x3x3 implicit-GEMM Conv3d, v_mfma_f32_32x32x2_f32)" if dtype == "f32"
                       else f"conv27_{dtype} (3x3x3 implicit-GEMM Conv3d: conv27_pp16 on v_mfma_f32_16x16x32_{dtype}; upsampled-input form, 4-wave forms and tail launches on v_mfma_f32_32x32x16_{dtype})"),
            # MFMA FLOPs actually issued (the structurally-zero z tap of the Z = 2 model is never staged nor multiplied:
            # 2/3 of the dense-conv count) / launch duration, against the dense peak: <= 1 by construction
            "achieved": round(executed, 3), "peak": peak, "unit": "TFLOP/s", "frac": round(executed / peak, 4),
            "traffic": traffic, "traffic_source": traffic_src,
            "launches_timed": prof["launches"], "avg_launch_ms": round(ms / n, 4),
            "executed_gflop_per_launch": round(prof["executed_flops"] / n / 1e9, 3),
            # the reference's own count for the same launches (torch.utils.flop_counter: dense Conv3d incl. the zero tap)
            "achieved_nominal": round(nominal, 3), "nominal_gflop_per_launch": round(prof["nominal_flops"] / n / 1e9, 3),
            "alg_gbytes_per_s": round(prof["alg_bytes"] / (ms * 1e-3) / 1e9, 1) if ms else 0.0,
            "kernel_share_of_step_time": round(ms / (1e3 * dt_total), 4),
            # every kernel of the step: needed FLOPs per interior patch-step x 0.684 (z-aware) x patch-steps/s
            "whole_step_tflops": round(whole, 3), "whole_step_frac": round(whole / peak, 4)}


def run_sweep_bench(args, dev, sd, world, rank, steps, warmup):
    """The row-sharded ROI sweep (strong scaling).  Returns the `sweep` result dict (rank 0 fills the JSON from it)."""
    import torch
    from teramind_amd import launch
    from teramind_amd.brain import consistent_gene_provider, device_gene_provider
    from teramind_amd.config import PathConfig
    from teramind_amd.diffusion import SpacedDiffusionBeatGans
    from teramind_amd.unet import BeatGANsUNetModel
    cfg = PathConfig(gen_type="ddim", compute_dtype=args.sweep_dtype)
    log(f"sweep: packing {args.sweep_dtype} weights")
    model = BeatGANsUNetModel(cfg, dev).load_state_dict(sd)
    model.overlap_streams = args.overlap_streams
    T = 15                                                    # test_brn default: 15-step DDIM (test_brn.py:329-330)
    smp = SpacedDiffusionBeatGans(T, "ddim")
    # share_halo needs gene tiles that agree where they overlap (as tiles cut from one gene map do): the block-seeded provider
    share = bool(args.sweep_share_halo) and args.sweep_batch_tiles > 1
    genes = consistent_gene_provider(cfg, dev) if share else device_gene_provider(cfg, dev)

    def on_step(sw, s):
        if rank == 0:
            log(f"sweep: step {sw.epoch} done in {s:.2f} s")

    # warm-up steps run unprofiled; the hipEvent brackets cover exactly the timed steps
    res = launch.run_sweep(cfg, smp, model, genes, hnm=args.sweep_hnm, wnm=args.sweep_wnm, total_epochs=T, steps=steps,
                           warmup=warmup, device=dev, batch_tiles=args.sweep_batch_tiles, init="device", state="fp16",
                           on_step=on_step, after_warmup=lambda: model.profile(True), share_halo=share,
                           batch_rows=args.sweep_batch_rows if share else 1, prefetch_genes=not share,
                           cache_level0=bool(args.sweep_cache_level0))
    prof = model.profile_collect()
    model.profile(False)
    st = res["sweep"].local_state()
    assert torch.isfinite(st.float()).all(), "non-finite sweep state"
    # partition-independent digest of the whole ROI state: exact integer sums over the fp16 bit patterns of every rank's rows
    bits = st.contiguous().view(torch.int16).to(torch.int64) & 0xFFFF
    dig = torch.stack([bits.sum(), (bits * bits).sum(), torch.tensor(bits.numel(), device=bits.device)])
    if world > 1:
        import torch.distributed as dist
        dh = dig if dist.get_backend() == "nccl" else dig.cpu()
        dist.all_reduce(dh, op=dist.ReduceOp.SUM)
        dig = dh
    digest = [int(v) for v in dig.cpu()]
    tiles = args.sweep_hnm * args.sweep_wnm
    value = 400.0 * tiles * steps / res["dt"]
    # a whole T-step sweep = its first step (which also computes what the later steps reuse: level 0 of the RNA conditioning
    # with cache_level0, workspace set-up) + T - 1 steady steps
    first_s = launch.reduce_max([res["warmup_s"][0]], dev)[0] if res["warmup_s"] else None
    steady_s = res["dt"] / steps
    out = {"value": round(value, 3), "unit": "interior patch-steps/s", "scaling": "strong", "dtype": args.sweep_dtype,
           "workload": (f"fixed ROI of {args.sweep_hnm} x {args.sweep_wnm} test_brn tiles (256x256 px x 100 channels = 25 z-chunks x "
                        f"16 interior patches, P=4) per diffusion step, DDIM-15 schedule, {args.sweep_dtype} UNet arithmetic, one "
                        f"float16 state canvas per rank, tile rows split over {world} rank(s)"),
           "tiles": tiles, "steps": steps, "warmup": warmup, "s_per_step": round(res["dt"] / steps, 4),
           "tiles_per_model_call": args.sweep_batch_tiles * (args.sweep_batch_rows if share else 1),
           "window_tiles": [args.sweep_batch_rows if share else 1, args.sweep_batch_tiles],
           # one window per call: the patch columns neighbouring tiles share go through the encoder once (bit-identical for
           # gene tiles that agree where they overlap)
           "share_halo": share,
           # level 0 of the RNA conditioning (gene attention -> down_z) kept per model call from the first step of the sweep on (the
           # warm-up step here): the timed steps are steps 2 .. T of a sweep, the first one costs ~2.5 % more
           "cache_level0": bool(args.sweep_cache_level0), "overlap_streams": args.overlap_streams,
           "first_step_s": round(first_s, 4) if first_s is not None else None,
           "value_full_sweep": (round(400.0 * tiles * T / (first_s + (T - 1) * steady_s), 3) if first_s is not None else None),
           "value_full_sweep_note": f"patch-steps/s of a whole T = {T} sweep = 400 x tiles x T / (measured first step + (T - 1) x measured steady step)",
           "s_per_tile_step_per_gpu": round(res["dt"] / steps / max(1, -(-args.sweep_hnm // world) * args.sweep_wnm), 5),
           "world_size_rccl": res["world"], "backend": res["backend"],
           "exchange_ms_per_step": round(res["exchange_ms_per_step"], 3),
           "exchange_bytes_per_step_per_rank": res["exchange_bytes_per_step"],
           "weights_broadcast_bytes": int(model.arena().numel()) if world > 1 else 0,
           "rows_rank0": list(res["rows"]), "state_digest": digest}
    return out, prof, res["dt"], value, model, cfg


def tile_extra(model, cfg, dev, dtype, steps=3, warmup=1):
    """The stacked one-tile step in the sweep's arithmetic (BASELINE config 4's governing shape: 25 z-chunks x 5x5 padded
    patches, P = 4) on the model the sweep packed: pad+patchify -> UNet -> DDIM update, `steps` timed steps."""
    import torch
    from teramind_amd import synth
    from teramind_amd.diffusion import SpacedDiffusionBeatGans, pad_patchify, sampler_step
    b, P, C, ps = 25, 4, cfg.in_channels, cfg.patch_size
    smp = SpacedDiffusionBeatGans(T_STEPS, "ddim")
    ne = b * (P + 1) ** 2
    state = synth.normal("bench/tile/xT", (b, C, ps * P, ps * P), 7).to(dev)
    rna = synth.gene_counts("bench/tile/rna", (ne, cfg.gn_sz, cfg.gn_sz, cfg.rna_slc * 500), 7).to(dev)
    shape_only = torch.empty((b, C, ps * P, ps * P), device="meta")
    tmap = torch.tensor(smp.timestep_map, dtype=torch.int64, device=dev)

    def one_step(k, st):
        i = T_STEPS - 1 - k
        xp = pad_patchify(st, ps)
        eps = model(x=xp, t=tmap[i].expand(b).contiguous(), rna=rna, imgs=shape_only, patch_size=ps).pred
        return sampler_step(smp, i, xp, eps, None, b, P, P)

    for k in range(warmup):
        state = one_step(k, state)
    torch.cuda.synchronize()
    model.profile(True)
    t0 = time.perf_counter()
    for k in range(steps):
        state = one_step(warmup + k, state)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = model.profile_collect()
    model.profile(False)
    assert torch.isfinite(state).all(), "non-finite state"
    value = b * P * P * steps / dt
    # the same steps with the two halves of the z-chunks on two HIP streams (model.overlap_streams = 2: bit-identical results;
    # wall clock only -- per-kernel durations overlap, so the roofline above stays the one-stream figure)
    two = None
    if getattr(model, "overlap_streams", 1) == 1:
        model.overlap_streams = 2
        st2 = one_step(0, state)                                   # sizes the second half's workspace
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            st2 = one_step(k, st2)
        torch.cuda.synchronize()
        two = round(1e3 * (time.perf_counter() - t0) / steps, 3)
        model.overlap_streams = 1
    return {"workload": "one test_brn tile per step: 25 z-chunks x (5x5 padded -> 4x4 interior) patches (P=4, 625 padded + 400 collage "
                        "patches), DDIM T=50 schedule, mode B arithmetic, " + dtype,
            "dtype": dtype, "steps": steps, "warmup": warmup, "ms_per_step": round(1e3 * dt / steps, 3),
            "value": round(value, 3), "unit": "interior patch-steps/s",
            "ms_per_step_overlap_streams_2": two,
            "roofline": roofline_block(prof, dtype, dt, value, P, True)}


def worker(args):
    import torch
    import torch.distributed as dist
    import teramind_amd  # noqa: F401
    from teramind_amd import launch, synth
    from teramind_amd.config import PathConfig
    from teramind_amd.diffusion import SpacedDiffusionBeatGans, pad_patchify, sampler_step
    from teramind_amd.unet import BeatGANsUNetModel
    from teramind_amd.weights import hashed_state_dict

    rank, local_rank, world = launch.dist_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}")
    if args.rehearse:
        # every rank on cuda:0 over gloo: rehearses the multi-process path (launcher, partition, arena broadcast, strip
        # exchange, reductions) on a one-GPU box; the numbers it prints are not a measurement
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    launch.init_distributed("gloo" if args.rehearse else "nccl", dev)

    B_IMAGES, P, gen = (25, 4, "ddim") if args.tile else (32, 1, "ddpm")
    cfg = PathConfig(gen_type=gen, batch_size=B_IMAGES, compute_dtype=args.dtype)
    log("generating hashed weights")
    sd = hashed_state_dict(cfg, 0)
    out = None
    if not args.sweep:
        log("packing + uploading weights")
        model = BeatGANsUNetModel(cfg, dev).load_state_dict(sd)
        model.overlap_streams = args.overlap_streams
        log("model ready")
        launch.broadcast_arena(model)      # replaces DDP's construction-time parameter broadcast (test_brn.py:149)
        smp = SpacedDiffusionBeatGans(T_STEPS, gen)

        b, C, ps = B_IMAGES, cfg.in_channels, cfg.patch_size
        ne = b * (P + 1) ** 2
        seed = 100 + rank
        img = synth.normal("bench/xT", (b, C, ps * P, ps * P), seed).to(dev)
        rna = synth.gene_counts("bench/rna", (ne, cfg.gn_sz, cfg.gn_sz, cfg.rna_slc * 500), seed).to(dev)
        noise = [synth.normal(f"bench/nz{k}", (ne, C, ps, ps), seed).to(dev) for k in range(4)] if gen == "ddpm" else None
        shape_only = torch.empty((b, C, ps * P, ps * P), device="meta")
        tmap = torch.tensor(smp.timestep_map, dtype=torch.int64, device=dev)

        pyr = {}

        def one_step(k, state):
            i = T_STEPS - 1 - (k % T_STEPS)
            xp = pad_patchify(state, ps)
            t = tmap[i].expand(b).contiguous()
            cond = rna
            if not args.tile:
                # mode A (configs[1]: the 50-step sample loop of SpacedDiffusionBeatGans.sample): the RNA conditioning
                # pyramid is computed once per sample of T = 50 steps, as the sampler does it; the first step of the timed
                # region recomputes it, so the K timed steps carry one such computation (50 / K times its real share)
                if k % T_STEPS == 0 or k == args.warmup or "p" not in pyr:
                    pyr["p"] = model.precompute_rna(rna, b, imgs=shape_only, patch_size=ps)
                cond = pyr["p"]
            eps = model(x=xp, t=t, rna=cond, imgs=shape_only, patch_size=ps).pred
            return sampler_step(smp, i, xp, eps, noise[k % 4] if gen == "ddpm" else None, b, P, P)

        state = img
        log("inputs resident; warm-up")
        for k in range(args.warmup):
            state = one_step(k, state)
            torch.cuda.synchronize()
            log(f"warm-up step {k} done")
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        model.profile(True)
        t0 = time.perf_counter()
        for k in range(args.steps):
            state = one_step(args.warmup + k, state)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        prof = model.profile_collect()
        model.profile(False)
        log(f"timed region done: {dt:.3f} s for {args.steps} steps")
        if world > 1:
            dist.barrier()
        dt = launch.reduce_max([dt], dev)[0]
        assert torch.isfinite(state).all(), "non-finite state"
        del model
        if rank == 0:
            units = b * P * P * args.steps * world
            value = units / dt
            out = {
                "metric": "denoising steps/sec on 64x64x(2 stains x 2 z) patches (interior patch-steps/s)",
                "value": round(value, 3), "unit": "interior patch-steps/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
                "data": "synthetic (hashed weights seed 0, N(0,1) state/noise, sparse integer gene counts)",
                "config": {"workload": ("configs[1]: b=32 images x 1 interior 64x64 patch (P=1, 128 padded + 32 collage "
                                        "patches/step), rna_slc=4, rna_num=229, stain=all, DDPM T=50 schedule, " + args.dtype + ", "
                                        "mode A (pad+patchify -> UNet -> DDPM update)") if not args.tile else
                                       ("one test_brn tile per step: 25 z-chunks x (5x5 padded -> 4x4 interior) patches "
                                        "(P=4, 625 padded + 400 collage patches), DDIM T=50 schedule, mode B arithmetic, " + args.dtype),
                           "per_gpu_patches_per_step": b * P * P, "parallelism": f"dp{world} (independent patch batches)"},
                "full_50_step_patches_per_s": round(value / T_STEPS, 4),
                "roofline": roofline_block(prof, args.dtype, dt, value / world, P, args.tile),
            }
    sweep_steps, sweep_warm = (args.steps, args.warmup) if args.sweep else (args.sweep_steps, 1)
    if args.sweep or not args.no_sweep:
        torch.cuda.empty_cache()
        sw_out, sw_prof, sw_dt, sw_value, sw_model, sw_cfg = run_sweep_bench(args, dev, sd, world, rank, sweep_steps, sweep_warm)
        # the 16-bit roofline belongs in the driver's line: the sweep's own conv27 figures, and the stacked one-tile step
        # (config 4's governing shape) on the same packed model
        sw_roof = roofline_block(sw_prof, args.sweep_dtype, sw_dt, sw_value / world, 4, True, sweep=True)
        extra = None
        if not args.sweep and not args.no_tile_extra:
            torch.cuda.empty_cache()
            extra = tile_extra(sw_model, sw_cfg, dev, args.sweep_dtype)
        del sw_model
        if rank == 0 and args.sweep:
            out = {"metric": "denoising steps/sec on 64x64x(2 stains x 2 z) patches (interior patch-steps/s)",
                   "value": sw_out["value"], "unit": "interior patch-steps/s", "n_gpus": world, "steps": sweep_steps,
                   "warmup": sweep_warm, "ms_per_step": round(1e3 * sw_dt / sweep_steps, 3), "higher_is_better": True,
                   "scaling": "strong", "vs_baseline": None, "dtype": args.sweep_dtype,
                   "data": "synthetic (hashed weights seed 0, per-tile seeded N(0,1) initial state, sparse integer gene tiles)",
                   "config": {"workload": sw_out["workload"], "parallelism": f"row-sharded tile grid over {world} rank(s), "
                              "32-px halo-strip exchange per step (RCCL send/recv), weight arena broadcast once"},
                   "roofline": sw_roof,
                   "sweep": sw_out}
        elif rank == 0:
            sw_out["roofline"] = sw_roof
            out["sweep"] = sw_out
            if extra is not None:
                out["extra"] = {"tile_" + args.sweep_dtype: extra}
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:      # rank 0 at N=1 only (the host cores are shared at N>1)
            out["cpu_baseline"] = cpu_baseline(PathConfig(), sd)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", choices=["f32", "bf16", "f16"], default="f32",
                    help="f32 = BASELINE configs[1] (the headline); bf16 = config-4 arithmetic (bf16 3x3x3 convs)")
    ap.add_argument("--tile", action="store_true",
                    help="mode-B workload instead: one test_brn tile (25 z-chunks x 5x5 patches, P=4, DDIM) per step")
    ap.add_argument("--sweep", action="store_true",
                    help="measure ONLY the row-sharded tile sweep (strong scaling; --steps / --warmup are diffusion steps of the ROI)")
    ap.add_argument("--no-sweep", action="store_true", help="default mode: skip the short sweep appended as the `sweep` object")
    ap.add_argument("--overlap-streams", type=int, default=1,
                    help="2: the model runs the two halves of a call's images on two HIP streams (bit-identical; faster wall clock, but "
                         "per-kernel durations -- the roofline fields -- then include the other stream's interference). Default 1")
    ap.add_argument("--no-tile-extra", action="store_true", help="default mode: skip the stacked one-tile 16-bit step appended as `extra`")
    ap.add_argument("--sweep-hnm", type=int, default=8)
    ap.add_argument("--sweep-wnm", type=int, default=8)
    ap.add_argument("--sweep-steps", type=int, default=2, help="timed diffusion steps of the appended sweep (1 warm-up step)")
    ap.add_argument("--sweep-dtype", choices=["bf16", "f16", "f32"], default="bf16")
    ap.add_argument("--sweep-batch-tiles", type=int, default=8, help="tiles of a tile row per model call")
    ap.add_argument("--sweep-batch-rows", type=int, default=2, help="tile rows per model call (shared-halo windows only)")
    ap.add_argument("--sweep-cache-level0", type=int, default=1,
                    help="1: keep level 0 of the RNA conditioning of every model call across the diffusion steps (59 KB per patch)")
    ap.add_argument("--sweep-share-halo", type=int, default=1,
                    help="1: the tiles of a call form one window (shared encoder patch columns computed once); 0: stacked tiles")
    ap.add_argument("--rehearse", action="store_true",
                    help="N > 1 on a one-GPU box: every rank on cuda:0, gloo instead of RCCL (checks the plumbing, measures nothing)")
    args = ap.parse_args()
    from teramind_amd import launch          # imports neither torch nor the HIP library
    if args.gpus > 1 and not launch.launched_as_rank():
        # one fresh process per GPU, started before anything in this process touches a GPU (test_brn.py:349-351 mp.spawn)
        sys.exit(launch.spawn_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))
    worker(args)


if __name__ == "__main__":
    main()
