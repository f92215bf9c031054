"""-m gpu: the training slice -- one ResBlock's training-mode forward (dropout with a supplied
keep mask) and backward on the HIP kernels, gradient-checked against torch.autograd over the oracle's functional blocks
(oracle/teramind_cpu.py) on the same parameters, inputs and mask; one AttnBlock (gene cross-attention) forward and backward
against the REFERENCE module's own autograd (tests/golden/train_attn_ref.npz, oracle/make_train_block_golden.py)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import util
from oracle import teramind_cpu as tc
from teramind_amd.training import AttnBlockTrain, ResBlockTrain
from train_cases import ATTN_CASES, make_attn_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref_block(P, x, scale, shift, mask, p_drop, per_image):
    """ResBlock._forward in training mode (model/MBAblocks.py:237-299, apply_conditions :302-368) on the oracle's pieces."""
    img = torch.arange(x.shape[0]) // per_image
    h = tc.silu(tc.rms_norm_channels(x, P["in_layers.0.weight"]))
    h = F.conv3d(h, P["in_layers.2.weight"], P["in_layers.2.bias"], padding=1)
    h = tc.rms_norm_channels(h, P["out_layers.0.weight"])
    h = h * (1 + scale[img][:, :, None, None, None]) + shift[img][:, :, None, None, None]
    h = tc.silu(h)
    if mask is not None:
        h = h * mask / (1.0 - p_drop)
    h = F.conv3d(h, P["out_layers.3.weight"], P["out_layers.3.bias"], padding=1)
    sk = F.conv3d(x, P["skip_connection.weight"], P["skip_connection.bias"]) if "skip_connection.weight" in P else x
    return sk + h


@pytest.mark.parametrize("N,Cin,Cout,S,per_image,drop", [(2, 24, 16, 8, 1, True), (4, 16, 16, 8, 2, True), (3, 13, 40, 16, 1, False),
                                                         (2, 96, 64, 16, 2, True)])
def test_resblock_training_forward_and_gradients(N, Cin, Cout, S, per_image, drop):
    g = torch.Generator().manual_seed(5)
    r = lambda *s, k=1.0: torch.randn(*s, generator=g) * k
    P = {"in_layers.0.weight": (torch.rand(1, Cin, 1, 1, generator=g) + 0.5),
         "in_layers.2.weight": r(Cout, Cin, 3, 3, 3, k=(Cin * 27) ** -0.5), "in_layers.2.bias": r(Cout, k=0.1),
         "out_layers.0.weight": (torch.rand(1, Cout, 1, 1, generator=g) + 0.5),
         "out_layers.3.weight": r(Cout, Cout, 3, 3, 3, k=(Cout * 27) ** -0.5), "out_layers.3.bias": r(Cout, k=0.1)}
    if Cin != Cout:
        P["skip_connection.weight"] = r(Cout, Cin, 1, 1, 1, k=Cin ** -0.5)
        P["skip_connection.bias"] = r(Cout, k=0.1)
    nimg = (N + per_image - 1) // per_image
    x = r(N, Cin, 2, S, S)
    scale, shift = r(nimg, Cout, k=0.3), r(nimg, Cout, k=0.3)
    p_drop = 0.1
    mask = (torch.rand(N, Cout, 2, S, S, generator=g) > p_drop).float() if drop else None
    dout = r(N, Cout, 2, S, S)
    # reference: autograd over the oracle's functional pieces
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    xr, scr, shr = x.clone().requires_grad_(True), scale.clone().requires_grad_(True), shift.clone().requires_grad_(True)
    ref = _ref_block(Pr, xr, scr, shr, mask, p_drop, per_image)
    ref.backward(dout)
    # HIP
    blk = ResBlockTrain(P, DEV)
    out = blk.forward(x.to(DEV), scale, shift, mask, p_drop, per_image)
    dx, dscale, dshift, grads = blk.backward(dout.to(DEV))
    tol = lambda t: 2e-5 * max(1.0, float(t.abs().max()))
    assert torch.allclose(out.cpu(), ref.detach(), atol=tol(ref), rtol=1e-5), util.report("forward", out, ref.detach())
    assert torch.allclose(dx.cpu(), xr.grad, atol=tol(xr.grad), rtol=1e-4), util.report("dx", dx, xr.grad)
    assert torch.allclose(dscale, scr.grad, atol=1e-3 * max(1.0, float(scr.grad.abs().max())), rtol=1e-4), util.report("dscale", dscale, scr.grad)
    assert torch.allclose(dshift, shr.grad, atol=1e-3 * max(1.0, float(shr.grad.abs().max())), rtol=1e-4), util.report("dshift", dshift, shr.grad)
    for k, gr in grads.items():
        rg = Pr[k].grad
        assert gr.shape == rg.shape, k
        # sums over N * 2 * S * S voxels of O(1) terms in fp32, different summation order (two-stage, reproducible: no float atomics;
        # a second run must give the same bits): relative to the gradient's scale
        assert torch.allclose(gr, rg, atol=2e-4 * max(1.0, float(rg.abs().max())), rtol=1e-3), (k, util.report(k, gr, rg))
    # reproducible gradients: the per-channel sums are two-stage reductions in a fixed order (no float atomics)
    blk.forward(x.to(DEV), scale, shift, mask, p_drop, per_image)
    dx2, dscale2, dshift2, grads2 = blk.backward(dout.to(DEV))
    assert torch.equal(dx2, dx) and torch.equal(dscale2, dscale) and torch.equal(dshift2, dshift)
    assert all(torch.equal(grads2[k], grads[k]) for k in grads)


def _rel(a, b):
    a, b = a.detach().cpu().double(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("name", sorted(ATTN_CASES))
def test_attn_block_forward_and_gradients_vs_reference_autograd(name):
    """AttnBlock._forward with cond (MBAblocks.py:480-489): output, dL/dx, dL/dcond and all 18 parameter gradients for
    L = sum(out * dout) against the reference module run in float64 on CPU.  The Linears run on the fp32 MFMA conv path
    (error-compensated bf16 splits, ~1e-6 relative), so the bound is a relative L2 error of 1e-4 per tensor."""
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "train_attn_ref.npz"))
    x, cond, dout, params = make_attn_inputs(name)
    blk = AttnBlockTrain(params, DEV)
    out = blk.forward(x.to(DEV), cond.to(DEV))
    assert _rel(out, gold[f"{name}/out"]) < 1e-4, util.report("out", out, torch.as_tensor(gold[f"{name}/out"]))
    dx, dcond, grads = blk.backward(dout.to(DEV))
    assert _rel(dx, gold[f"{name}/dx"]) < 1e-4, util.report("dx", dx, torch.as_tensor(gold[f"{name}/dx"]))
    assert _rel(dcond, gold[f"{name}/dcond"]) < 1e-4, util.report("dcond", dcond, torch.as_tensor(gold[f"{name}/dcond"]))
    keys = sorted(k[len(name) + 6:] for k in gold.files if k.startswith(f"{name}/grad/"))
    assert keys == sorted(params) == sorted(grads), (keys, sorted(grads))
    for k in keys:
        ref = torch.as_tensor(gold[f"{name}/grad/{k}"])
        assert grads[k].shape == ref.shape and _rel(grads[k], ref) < 1e-4, (k, util.report(k, grads[k], ref))
    # reproducible: a second run gives the same bits (two-stage reductions, no float atomics)
    blk.forward(x.to(DEV), cond.to(DEV))
    dx2, dcond2, grads2 = blk.backward(dout.to(DEV))
    assert torch.equal(dx2, dx) and torch.equal(dcond2, dcond) and all(torch.equal(grads2[k], grads[k]) for k in grads)
