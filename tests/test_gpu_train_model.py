"""-m gpu: the whole-model training slice (teramind_amd.train_model).  The forward of the tape equals
the product inference forward (pred and pred2); the loss and the gradient of ALL 403 parameter tensors equal the reference's
own `training_losses(...).backward()` on the same seeded inputs (tests/golden/train_grad_ref.npz, minted by
oracle/make_train_grad_golden.py from /root/reference in float32 on CPU)."""
import os

import numpy as np
import pytest
import torch

from teramind_amd.config import PathConfig
from teramind_amd.diffusion import SpacedDiffusionBeatGans
from teramind_amd.train_model import AdamTrainer, UNetTrain, training_loss_and_grads
from teramind_amd.weights import hashed_state_dict
from train_cases import GRAD_CASES, GRAD_CFG, GRAD_FULL_MAX, GRAD_PROBES, grad_probe, make_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_training_forward_equals_inference_forward():
    from teramind_amd.unet import BeatGANsUNetModel
    cfg = PathConfig(net_ch=64, rna_num=37)            # the inference engine takes net_ch in multiples of 64
    sd = hashed_state_dict(cfg, 0)
    b, ps = 2, cfg.patch_size
    g = torch.Generator().manual_seed(4)
    x = torch.randn((b * 4, cfg.in_channels, ps, ps), generator=g)
    rna = (torch.rand((b * 4, cfg.gn_sz, cfg.gn_sz, cfg.rna_slc * 500), generator=g) < 0.02).float() * 3.0
    t = torch.tensor([17, 803])
    net = UNetTrain(cfg, sd, DEV)
    pred, pred2 = net.forward(x, t, rna, b)
    model = BeatGANsUNetModel(cfg, device=DEV)
    model.load_state_dict(sd, strict=True)
    ref = model(x=x.to(DEV), t=t.to(DEV), rna=rna.to(DEV), imgs=torch.empty((b, cfg.in_channels, ps, ps), device="meta"), patch_size=ps,
                want_pred2=True)
    for a, r, nm in ((pred, ref.pred, "pred"), (pred2, ref.pred2, "pred2")):
        err = float((a - r).norm() / r.norm())
        assert a.shape == r.shape and err < 1e-4, (nm, err)


@pytest.mark.parametrize("name", sorted(GRAD_CASES))
def test_whole_model_gradients_vs_reference_backward(name):
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "train_grad_ref.npz"))
    seed, loss_type, crop = GRAD_CASES[name]
    cfg = PathConfig(**GRAD_CFG)
    sd = hashed_state_dict(cfg, 0)
    x_pad, rna, imgs, t, pos, mask, idx, noise = make_inputs(seed)
    net = UNetTrain(cfg, sd, DEV)
    sampler = SpacedDiffusionBeatGans(1000, "ddpm")
    loss, grads = training_loss_and_grads(net, sampler, x_pad, rna, t, mask, noise, crop, cfg.patch_size, loss_type)
    ref_loss = float(gold[f"{name}/loss"])
    assert abs(loss - ref_loss) <= 2e-5 * abs(ref_loss), (loss, ref_loss)
    keys = sorted(k[len(name) + 6:] for k in gold.files if k.startswith(f"{name}/norm/"))
    assert keys == sorted(sd) and sorted(grads) == keys, (set(keys) ^ set(grads))
    bad = []
    for k in keys:
        g = grads[k].double().reshape(-1).numpy()
        nref = float(gold[f"{name}/norm/{k}"])
        # a gradient with relative error eps moves a projection onto a uniform[-1, 1) probe by ~ eps |g| / sqrt(3); a wrong
        # gradient moves it by ~ |g| / sqrt(3): bounds 3e-3 |g| (projections) and 2e-3 (norm, whole small tensors)
        e_norm = abs(np.linalg.norm(g) - nref) / nref
        pr = np.array([float(g @ grad_probe(k, g.size, j)) for j in range(GRAD_PROBES)])
        e_proj = float(np.abs(pr - gold[f"{name}/proj/{k}"]).max()) / nref
        e_full = 0.0
        if g.size <= GRAD_FULL_MAX:
            rf = gold[f"{name}/full/{k}"].astype(np.float64).reshape(-1)
            e_full = float(np.linalg.norm(g - rf) / np.linalg.norm(rf))
        if not (e_norm < 2e-3 and e_proj < 3e-3 and e_full < 2e-3):
            bad.append((k, nref, e_norm, e_proj, e_full))
    assert not bad, f"{len(bad)} of {len(keys)} gradients off: " + "; ".join(f"{k} |g|={n:.3g} norm {a:.2e} proj {b:.2e} full {c:.2e}"
                                                                              for k, n, a, b, c in bad[:12])
    # reproducible: the same step again gives the same bits for every tensor (fixed-order reductions, no float atomics)
    loss2, grads2 = training_loss_and_grads(UNetTrain(cfg, sd, DEV), sampler, x_pad, rna, t, mask, noise, crop, cfg.patch_size, loss_type)
    assert loss2 == loss and all(torch.equal(grads2[k], grads[k]) for k in keys)


def test_optimizer_step_equals_torch_adam_with_clip():
    """AdamTrainer (tm_op_sumsq + tm_op_adam on the flat arena) against what the reference's loop calls -- torch.nn.utils.
    clip_grad_norm_(max_norm=1) and torch.optim.Adam(lr=2e-5, weight_decay=0) (experiment.py:207-219, 394-414) -- on the tiny
    model's parameters with the gradients of two real training steps (the second after the first update), plus one step with
    weight decay and two accumulated micro-batches."""
    cfg = PathConfig(**GRAD_CFG)
    sd = hashed_state_dict(cfg, 0)
    x_pad, rna, imgs, t, pos, mask, idx, noise = make_inputs(3)
    net = UNetTrain(cfg, sd, DEV)
    sampler = SpacedDiffusionBeatGans(1000, "ddpm")
    opt = AdamTrainer(net)
    ref_p = [torch.nn.Parameter(v.clone().float()) for v in sd.values()]
    ref_opt = torch.optim.Adam(ref_p, lr=2e-5, weight_decay=0)
    losses = []
    for step in range(2):
        loss, grads = training_loss_and_grads(net, sampler, x_pad, rna, t, mask, noise, (1, 0), cfg.patch_size, "mse")
        losses.append(loss)
        opt.accumulate(grads)
        info = opt.step()
        for p_, k in zip(ref_p, sd):
            p_.grad = grads[k].clone().reshape(p_.shape)
        tn = torch.nn.utils.clip_grad_norm_(ref_p, max_norm=1.0)
        ref_opt.step()
        assert abs(info["grad_norm"] - float(tn)) <= 1e-5 * float(tn) and info["clip_coef"] < 1.0
        for p_, k in zip(ref_p, sd):
            # |update| ~ lr = 2e-5 per step: agree to ~1 % of one update (an fp32 ulp of a parameter near 1 is 1.2e-7)
            assert torch.allclose(net.W[k], p_.detach(), rtol=0, atol=2.5e-7), (step, k, float((net.W[k] - p_.detach()).abs().max()))
    assert losses[1] < losses[0]                                   # the same batch again after one step
    # weight decay, two micro-batches, no clipping
    opt2 = AdamTrainer(net, lr=1e-3, weight_decay=0.01, grad_clip=0.0)
    ref_p2 = [torch.nn.Parameter(net.W[k].clone()) for k in sd]
    ref_opt2 = torch.optim.Adam(ref_p2, lr=1e-3, weight_decay=0.01)
    g2 = {k: 0.5 * v for k, v in grads.items()}
    opt2.accumulate(grads)
    opt2.accumulate(g2)
    opt2.step()
    for p_, k in zip(ref_p2, sd):
        p_.grad = (0.75 * grads[k]).reshape(p_.shape)
    ref_opt2.step()
    for p_, k in zip(ref_p2, sd):
        assert torch.allclose(net.W[k], p_.detach(), rtol=0, atol=1e-3 * 2e-3), (k, float((net.W[k] - p_.detach()).abs().max()))
