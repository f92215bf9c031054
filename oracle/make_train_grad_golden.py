#!/usr/bin/env python3
"""Mints tests/golden/train_grad_ref.npz FROM THE REFERENCE (TEST INFRASTRUCTURE): GaussianDiffusionBeatGans.training_losses
(diffusion/base.py:181-289) on the reference model built at the tiny configuration of tests/train_cases.py (GRAD_CFG), float32
(the reference's timestep embedding is float32 whatever the module dtype), CPU, eval mode (no dropout) with gradients, then `loss.backward()` through torch.autograd.  Stored per parameter: the
gradient's L2 norm, GRAD_PROBES projections onto seeded probe vectors (train_cases.grad_probe), and the whole gradient when the
tensor has at most GRAD_FULL_MAX elements; plus the loss and the two predictions.  The two `random.randrange` draws of the crop
are pinned by patching `random.randrange` for the call; the `device='cuda'` argument of base.py:224 is dropped as in
make_train_golden.py.  Run here:  python oracle/make_train_grad_golden.py"""
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
from train_cases import GRAD_CASES, GRAD_CFG, GRAD_FULL_MAX, GRAD_PROBES, grad_probe, make_inputs  # noqa: E402
from teramind_amd.config import PathConfig  # noqa: E402
from teramind_amd.weights import hashed_state_dict  # noqa: E402


def main():
    torch.set_num_threads(8)
    cfg = PathConfig(**GRAD_CFG)
    conf = rh.make_conf(nrna=cfg.rna_num, net_ch=cfg.net_ch)
    model = rh.make_model(conf)
    model.load_state_dict(hashed_state_dict(cfg, 0), strict=True)
    model.eval()
    out = {}
    for name, (seed, loss_type, (ix, iy)) in GRAD_CASES.items():
        sampler = rh.make_sampler(conf, 1000, "ddpm")
        from utils.choices import LossType
        sampler.loss_type = LossType.mse if loss_type == "mse" else LossType.l1
        x_pad, rna, imgs, t, pos, mask, idx, noise = make_inputs(seed)
        draws = [ix, iy]
        real_tensor, real_rr = torch.tensor, random.randrange
        torch.tensor = lambda *a, **k: real_tensor(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})
        random.randrange = lambda *a, **k: draws.pop(0)
        model.zero_grad()
        try:
            terms = sampler.training_losses(model=model, x_start=x_pad, r_start=(rna[0].clone(), rna[1].clone(), rna[2]),
                                            imgs=imgs, t=t, pos=pos, loss_mask=mask, idx=idx, patch_size=64, noise=noise)
        finally:
            torch.tensor, random.randrange = real_tensor, real_rr
        assert not draws
        loss = terms["loss"].mean()
        loss.backward()
        out[f"{name}/loss"] = np.array(float(loss), dtype=np.float64)
        for k, p in model.named_parameters():
            g = p.grad.detach().double().reshape(-1).numpy()
            out[f"{name}/norm/{k}"] = np.array(np.linalg.norm(g))
            out[f"{name}/proj/{k}"] = np.array([float(g @ grad_probe(k, g.size, j)) for j in range(GRAD_PROBES)])
            if g.size <= GRAD_FULL_MAX:
                out[f"{name}/full/{k}"] = g.astype(np.float32).reshape(p.shape)
        gn = {k: float(out[f"{name}/norm/{k}"]) for k, _ in model.named_parameters()}
        print(name, "loss", float(loss), "params", len(gn), "min/median/max grad norm", min(gn.values()), sorted(gn.values())[len(gn) // 2],
              max(gn.values()), flush=True)
        zero = [k for k, v in gn.items() if v == 0.0]
        print("zero-gradient tensors:", zero)
    p = os.path.join(ROOT, "tests", "golden", "train_grad_ref.npz")
    np.savez_compressed(p, **out)
    print("wrote", p, os.path.getsize(p), "bytes")


if __name__ == "__main__":
    main()
