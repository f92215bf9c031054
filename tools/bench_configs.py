#!/usr/bin/env python3
"""Row f4 measurement: one mode-A denoising step (b images x 1 interior patch; TM_DTYPE=f32|bf16|f16) for the other
configuration-surface points, same timing method as bench.py.  These paths are functional-first (generic
attention kernels, direct down_z); the table says what they cost, not that they are tuned.  One JSON line each."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import teramind_amd  # noqa: E402,F401
from teramind_amd import synth  # noqa: E402
from teramind_amd.config import PathConfig  # noqa: E402
from teramind_amd.diffusion import SpacedDiffusionBeatGans, pad_patchify, sampler_step  # noqa: E402
from teramind_amd.unet import BeatGANsUNetModel  # noqa: E402
from teramind_amd.weights import hashed_state_dict  # noqa: E402

dev = "cuda:0"
DTYPE = os.environ.get("TM_DTYPE", "f32")      # f32 | bf16 | f16
CASES = [(64, 4, "all", 229, 32), (64, 1, "all", 229, 32), (64, 8, "all", 229, 16), (64, 16, "all", 229, 8), (32, 4, "all", 229, 64),
         (128, 4, "all", 229, 8), (64, 4, "all", 500, 32), (64, 1, "all", 81, 32)]
if os.environ.get("TM_CASE"):
    CASES = [CASES[int(i)] for i in os.environ["TM_CASE"].split(",")]
for size, srna, stain, nrna, b in CASES:
    if DTYPE != "f32" and size == 32:
        continue
    cfg = PathConfig(patch_size=size, rna_slc=srna, stain=stain, rna_num=nrna, compute_dtype=DTYPE)
    model = BeatGANsUNetModel(cfg, dev).load_state_dict(hashed_state_dict(cfg, 0))
    smp = SpacedDiffusionBeatGans(50, "ddpm")
    C = cfg.in_channels
    img = synth.normal("bc/x", (b, C, size, size), 1).to(dev)
    rna = synth.gene_counts("bc/rna", (4 * b, cfg.gn_sz, cfg.gn_sz, srna * 500), 1).to(dev)
    nz = synth.normal("bc/nz", (4 * b, C, size, size), 2).to(dev)
    shape_only = torch.empty((b, C, size, size), device="meta")
    tmap = torch.tensor(smp.timestep_map, dtype=torch.int64, device=dev)

    def step(k, st):
        i = 49 - k
        xp = pad_patchify(st, size)
        eps = model(x=xp, t=tmap[i].expand(b).contiguous(), rna=rna, imgs=shape_only, patch_size=size).pred
        return sampler_step(smp, i, xp, eps, nz, b, 1, 1)
    st = img
    for k in range(2):
        st = step(k, st)
    torch.cuda.synchronize()
    model.profile(True)
    t0 = time.perf_counter()
    n = 5
    for k in range(n):
        st = step(2 + k, st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    pr = model.profile_collect()
    print(json.dumps({"dtype": DTYPE, "patch_size": size, "rna_slc": srna, "z_size": cfg.z_size, "stain": stain, "rna_num": nrna, "b": b,
                      "ms_per_step": round(dt * 1e3, 2), "interior_patch_steps_per_s": round(b / dt, 1),
                      "conv3x3_nominal_tflops": round(pr["nominal_flops"] / (pr["total_ms"] * 1e-3) / 1e12, 1) if pr["total_ms"] else None,
                      "conv3x3_executed_tflops": round(pr["executed_flops"] / (pr["total_ms"] * 1e-3) / 1e12, 1) if pr["total_ms"] else None,
                      "conv3x3_share": round(pr["total_ms"] / n / (dt * 1e3), 3)}), flush=True)
    del model
