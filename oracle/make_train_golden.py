#!/usr/bin/env python3
"""Mints tests/golden/train_loss.npz FROM THE REFERENCE (TEST INFRASTRUCTURE): GaussianDiffusionBeatGans.
training_losses (diffusion/base.py:181-289) evaluated on CPU with the model in eval() mode, seeded python
`random` (the 2x2 crop position), given noise, hashed weights.  The reference builds `index` with device='cuda'
(base.py:224); on this CPU-only container torch.tensor is wrapped for the duration of the call to drop that
argument -- nothing else is touched.  Run here:  python oracle/make_train_golden.py"""
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import teramind_amd  # noqa: E402,F401
from oracle import ref_harness as rh  # noqa: E402
from train_cases import CASES, make_inputs  # noqa: E402
from teramind_amd.config import PathConfig  # noqa: E402
from teramind_amd.weights import hashed_state_dict  # noqa: E402


def main():
    torch.set_num_threads(8)
    cfg = PathConfig()
    conf = rh.make_conf()
    model = rh.make_model(conf)
    model.load_state_dict(hashed_state_dict(cfg, 0), strict=True)
    model.eval()
    out = {}
    for name, (seed, loss_type) in CASES.items():
        sampler = rh.make_sampler(conf, 1000, "ddpm")
        from utils.choices import LossType
        sampler.loss_type = LossType.mse if loss_type == "mse" else LossType.l1
        x_pad, rna, imgs, t, pos, mask, idx, noise = make_inputs(seed)
        random.seed(seed)
        ix, iy = random.randrange(pos.shape[0] - 1), random.randrange(pos.shape[1] - 1)
        random.seed(seed)
        real_tensor = torch.tensor
        torch.tensor = lambda *a, **k: real_tensor(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})
        try:
            with torch.inference_mode():
                terms = sampler.training_losses(model=model, x_start=x_pad, r_start=(rna[0].clone(), rna[1].clone(), rna[2]),
                                                imgs=imgs, t=t, pos=pos, loss_mask=mask, idx=idx, patch_size=64, noise=noise)
        finally:
            torch.tensor = real_tensor
        out[f"{name}/loss"] = np.array(float(terms["loss"]), dtype=np.float64)
        out[f"{name}/crop"] = np.array([ix, iy])
        xt = terms["x_t"]
        out[f"{name}/x_t_stats"] = np.array([xt.double().mean().item(), xt.double().abs().max().item(), xt.double().std().item()])
        print(name, "loss", float(terms["loss"]), "crop", ix, iy, flush=True)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "train_loss.npz"), **out)


if __name__ == "__main__":
    main()
