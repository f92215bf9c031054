#!/usr/bin/env python3
"""Headline benchmark of the denoising hot path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one denoising step of the hot path over one batch of synthetic input:
pad+patchify -> UNet (tm_unet_forward) -> DDPM update (tm_sampler_step) for b=32 images of one
64x64 interior patch each (BASELINE configs[1]: batch_size=32, patch_size=64, rna_slc=4, T=50
DDPM, fp32; mode A, P=1 => 128 padded encoder patches, 32 collage decoder patches per step).
All inputs are resident in HBM before the timed region.  The unit is the interior patch-step
(SURVEY.md section 8d).  With N > 1 every rank runs its own batch of 32 patches (the path
shards by independent patches: weak scaling, no data-path collective); the only collectives
are the one-off RCCL broadcast of rank 0's packed weight arena and the timing barrier.

Prints ONE JSON line on rank 0, with `roofline` (dominant kernel, the 3x3x3 implicit-GEMM conv: nominal dense-conv
FLOPs per launch / average launch duration from hipEvents recorded on the launch stream during
the timed region, against the fp32 MFMA peak) and `cpu_baseline` (oracle/teramind_cpu.py on the
host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0    # same table, "Peak BF16/FP16 MFMA ~2.5 PF dense"
B_IMAGES, P, T_STEPS = 32, 1, 50
NEEDED_GFLOP_PER_PATCH_STEP = 320.1   # SURVEY.md section 8(d), P=1 (whole path, 2*MAC)


_T0 = time.perf_counter()


def log(msg):
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(cfg, sd, budget_steps=6, b=8):
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
    ap.add_argument("--pmc-json", default=None,
                    help="rocprofv3 --pmc derived HBM bytes per launch of the dominant conv kernel (default: "
                         "profiles/conv27_traffic[_bf16][_tile].json for the selected workload)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import teramind_amd  # noqa: F401
    from teramind_amd import synth
    from teramind_amd.config import PathConfig
    from teramind_amd.diffusion import SpacedDiffusionBeatGans, pad_patchify, sampler_step
    from teramind_amd.unet import BeatGANsUNetModel
    from teramind_amd.weights import hashed_state_dict

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("for --gpus N > 1 launch with torch.distributed.run --nproc-per-node N")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    global B_IMAGES, P
    gen = "ddpm"
    if args.tile:
        B_IMAGES, P, gen = 25, 4, "ddim"
    cfg = PathConfig(gen_type=gen, batch_size=B_IMAGES, compute_dtype=args.dtype)
    log("generating hashed weights")
    sd = hashed_state_dict(cfg, 0)
    log("packing + uploading weights")
    model = BeatGANsUNetModel(cfg, dev).load_state_dict(sd)
    log("model ready")
    if world > 1:      # replaces DDP's construction-time parameter broadcast (test_brn.py:149)
        dist.broadcast(model.arena(), src=0)
    smp = SpacedDiffusionBeatGans(T_STEPS, gen)

    b, C, ps = B_IMAGES, cfg.in_channels, cfg.patch_size
    ne = b * (P + 1) ** 2
    seed = 100 + rank
    img = synth.normal("bench/xT", (b, C, ps * P, ps * P), seed).to(dev)
    rna = synth.gene_counts("bench/rna", (ne, cfg.gn_sz, cfg.gn_sz, cfg.rna_slc * 500), seed).to(dev)
    noise = [synth.normal(f"bench/nz{k}", (ne, C, ps, ps), seed).to(dev) for k in range(4)] if gen == "ddpm" else None
    shape_only = torch.empty((b, C, ps * P, ps * P), device="meta")
    tmap = torch.tensor(smp.timestep_map, dtype=torch.int64, device=dev)

    def one_step(k, state):
        i = T_STEPS - 1 - (k % T_STEPS)
        xp = pad_patchify(state, ps)
        t = tmap[i].expand(b).contiguous()
        eps = model(x=xp, t=t, rna=rna, imgs=shape_only, patch_size=ps).pred
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
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(state).all(), "non-finite state"

    if rank == 0:
        units = b * P * P * args.steps * world
        value = units / dt
        avg_ms = prof["total_ms"] / max(1, prof["launches"])
        achieved = prof["nominal_flops"] / (prof["total_ms"] * 1e-3) / 1e12 if prof["total_ms"] else 0.0
        traffic = None
        peak = PEAK_F32_MFMA_TFLOPS if args.dtype == "f32" else PEAK_BF16_MFMA_TFLOPS
        pmc_json = args.pmc_json or os.path.join(ROOT, "profiles", "conv27_traffic" + ("_bf16" if args.dtype != "f32" else "") +
                                                 ("_tile" if args.tile else "") + ".json")
        if os.path.exists(pmc_json):
            try:
                traffic = json.load(open(pmc_json)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
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
                                    "(P=4, 625 padded + 400 collage patches), DDIM T=50 schedule, mode B arithmetic"),
                       "per_gpu_patches_per_step": b * P * P, "parallelism": f"dp{world} (independent patch batches)"},
            "full_50_step_patches_per_s": round(value / T_STEPS, 4),
            "roofline": {"bound": "mfma", "kernel": ("conv3d_mfma<2,*,*> (3x3x3 implicit-GEMM Conv3d, v_mfma_f32_32x32x2_f32)" if args.dtype == "f32"
                                                      else f"conv27_{args.dtype} (3x3x3 implicit-GEMM Conv3d, v_mfma_f32_32x32x16_{args.dtype})"),
                         "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic,
                         "launches_timed": prof["launches"], "launches_per_step": prof["launches"] // max(1, args.steps),
                         "avg_launch_ms": round(avg_ms, 4),
                         "nominal_gflop_per_launch": round(prof["nominal_flops"] / max(1, prof["launches"]) / 1e9, 3),
                         "executed_mfma_tflops": round(prof["executed_flops"] / (prof["total_ms"] * 1e-3) / 1e12, 3) if prof["total_ms"] else 0.0,
                         # MFMA instructions actually issued (structurally-zero z taps skipped) over the dense peak: the
                         # pipe-utilisation figure; `frac` prices the reference's nominal FLOPs and can exceed 1
                         "frac_executed": round(prof["executed_flops"] / (prof["total_ms"] * 1e-3) / 1e12 / peak, 4) if prof["total_ms"] else 0.0,
                         "alg_gbytes_per_s": round(prof["alg_bytes"] / (prof["total_ms"] * 1e-3) / 1e9, 1) if prof["total_ms"] else 0.0,
                         "conv27_share_of_step_time": round(prof["total_ms"] / (1e3 * dt), 4),
                         "whole_step_needed_tflops": round((NEEDED_GFLOP_PER_PATCH_STEP if P == 1 else 202.6) * value / world / 1e3, 3)},
        }
        if not args.no_cpu_baseline and world == 1:      # rank 0 at N=1 only (the host cores are shared at N>1)
            out["cpu_baseline"] = cpu_baseline(cfg, sd)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
