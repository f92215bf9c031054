"""-m gpu: sampler kernels (tm_sampler_step, tm_pad_patchify) and the sample() loop vs oracle."""
import pytest
import torch
import torch.nn.functional as F

import util
from oracle import teramind_cpu as tc
from teramind_amd import synth
from teramind_amd.config import PathConfig
from teramind_amd.diffusion import SpacedDiffusionBeatGans, pad_patchify, sampler_step

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("b,P1,P2", [(1, 1, 1), (2, 2, 3)])
def test_pad_patchify_bit_exact(b, P1, P2):
    img = synth.normal("img", (b, 4, 64 * P1, 64 * P2), 2)
    ref = tc.patchify(F.pad(img, (32, 32, 32, 32)), 64)
    got = pad_patchify(img.to(DEV), 64)
    assert torch.equal(got.cpu(), ref)


@pytest.mark.parametrize("gen,T,i", [("ddpm", 50, 49), ("ddpm", 50, 0), ("ddim", 15, 7), ("ddim", 15, 0), ("ddim", 50, 31)])
@pytest.mark.parametrize("b,P1,P2", [(1, 1, 1), (2, 2, 3)])
def test_sampler_step_bit_exact(gen, T, i, b, P1, P2):
    """Elementwise float32 math in the reference's operation order: bit-exact vs the oracle."""
    sch = tc.make_schedule(T, gen)
    smp = SpacedDiffusionBeatGans(T, gen)
    ne = b * (P1 + 1) * (P2 + 1)
    xp = synth.normal("xp", (ne, 4, 64, 64), 1)
    eps = synth.normal("eps", (b * P1 * P2, 4, 64, 64), 2)
    noise = synth.normal("nz", (ne, 4, 64, 64), 3) if gen == "ddpm" else None
    ref = tc.sampler_step(sch, gen, xp, eps, i, P1, P2, noise)
    got = sampler_step(smp, i, xp.to(DEV), eps.to(DEV), None if noise is None else noise.to(DEV), b, P1, P2)
    assert torch.equal(got.cpu(), ref), util.report("step", got, ref)
