"""-m gpu: the other configuration-surface points -- patch_size 32 / 128, rna_slc 1 / 8 / 16
(z_size 1 / 4 / 8: general-Z conv form, down_z kernels 1 / 5 / 9, window sizes 8 ... 512), single stains, the 500-gene
mice and the 81-gene M2H subset -- HIP path vs the CPU oracle and vs outputs minted from the real reference."""
import os

import numpy as np
import pytest
import torch

import config_cases as cc
import util
from oracle import teramind_cpu as tc
from teramind_amd.unet import BeatGANsUNetModel, GeneAttnModel
from teramind_amd.weights import hashed_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("case", cc.CONFIGS, ids=cc.tag_of)
def test_unet_other_configs_vs_oracle_and_reference(case):
    cfg = cc.path_config(case)
    sd = hashed_state_dict(cfg, 0)
    x, rna, t = cc.inputs(cfg)
    torch.set_num_threads(16)
    with torch.inference_mode():
        ref, ref2 = tc.unet_forward(sd, tc.oracle_config_from(cfg), x, t, rna, 2, 2, want_pred2=True)
    m = BeatGANsUNetModel(cfg, DEV).load_state_dict(sd)
    out = m(x=x.to(DEV), t=t.to(DEV), rna=rna.to(DEV), imgs=torch.zeros(1, cfg.in_channels, cfg.patch_size, cfg.patch_size),
            patch_size=cfg.patch_size, want_pred2=True)
    assert torch.allclose(out.pred.cpu(), ref, atol=2e-4, rtol=0), util.report("pred", out.pred, ref)
    assert torch.allclose(out.pred2.cpu(), ref2, atol=2e-4, rtol=0), util.report("pred2", out.pred2, ref2)
    gold = np.load(os.path.join(util.GOLDEN, "unet_configs.npz"))
    for name, got in (("pred", out.pred), ("pred2", out.pred2)):
        assert np.abs(cc.digest(got) - gold[f"{cc.tag_of(case)}/{name}"]).max() < 2e-4, (case, name)


def test_unet_other_config_with_interior_grid():
    """P = 2 x 3 interior patches on a z_size-4 config (general-Z conv + collage + generic attention windows)."""
    from teramind_amd import synth
    cfg = cc.path_config((64, 8, "all", 229))
    sd = hashed_state_dict(cfg, 0)
    b, p1, p2 = 1, 3, 4
    ne = b * p1 * p2
    x = synth.normal("cfg/x23", (ne, cfg.in_channels, 64, 64), 2)
    rna = synth.gene_counts("cfg/rna23", (ne, 4, 4, 8 * 500), 2)
    t = torch.tensor([77])
    with torch.inference_mode():
        ref, _ = tc.unet_forward(sd, tc.oracle_config_from(cfg), x, t, rna, p1, p2)
    out = BeatGANsUNetModel(cfg, DEV).load_state_dict(sd)(x=x.to(DEV), t=t.to(DEV), rna=rna.to(DEV),
                                                          imgs=torch.zeros(b, cfg.in_channels, 128, 192), patch_size=64)
    assert torch.allclose(out.pred.cpu(), ref, atol=2e-4, rtol=0), util.report("pred", out.pred, ref)


def test_attention_maps_500_genes():
    """The attention-map model with the 500-gene panel (generic gene-attention kernel, G > 232)."""
    from teramind_amd import synth
    cfg = cc.path_config((64, 4, "all", 500))
    sd = hashed_state_dict(cfg, 0, vis_only=True)
    rna = synth.gene_counts("cfg/rna_vis", (3, 4, 4, 2000), 1, density=0.05)
    attn, mid = GeneAttnModel(cfg, DEV).load_state_dict(sd, strict=False)(rna=rna.to(DEV))
    with torch.inference_mode():
        ra, rm = tc.gene_attention_maps(sd, tc.oracle_config_from(cfg), rna)
    assert attn.shape == (4, 3, 500, 500) and torch.allclose(attn.cpu(), ra, atol=2e-6, rtol=1e-4)
    assert torch.equal(mid.cpu(), rm)


@pytest.mark.parametrize("dtype,rel_tol", [("bf16", 7e-3), ("f16", 9e-4)])
@pytest.mark.parametrize("case", [(64, 1, "all", 229), (64, 8, "all", 229), (128, 4, "DAPI", 229), (64, 16, "all", 229),
                                  (64, 4, "all", 500), (64, 1, "all", 81)], ids=cc.tag_of)
def test_unet_other_configs_16bit_modes(case, dtype, rel_tol):
    """The 16-bit operand modes on the other z sizes (the 16-bit conv stages only the input planes that exist: 1, 2 or 3
    per output plane), patch_size 128 and the generic gene / window attention fall-backs: relative L2 vs the fp32 oracle
    within the operand-rounding bound of the type."""
    cfg = cc.path_config(case, compute_dtype=dtype)
    sd = hashed_state_dict(cfg, 0)
    x, rna, t = cc.inputs(cfg)
    torch.set_num_threads(16)
    with torch.inference_mode():
        ref, ref2 = tc.unet_forward(sd, tc.oracle_config_from(cfg), x, t, rna, 2, 2, want_pred2=True)
    m = BeatGANsUNetModel(cfg, DEV).load_state_dict(sd)
    out = m(x=x.to(DEV), t=t.to(DEV), rna=rna.to(DEV), imgs=torch.zeros(1, cfg.in_channels, cfg.patch_size, cfg.patch_size),
            patch_size=cfg.patch_size, want_pred2=True)
    for got, want in ((out.pred.cpu(), ref), (out.pred2.cpu(), ref2)):
        rel = ((got - want).norm() / want.norm()).item()
        assert rel < rel_tol, (case, dtype, rel)


def test_unsupported_configs_fail_loudly():
    with pytest.raises(RuntimeError, match="gene-token width"):
        BeatGANsUNetModel(cc.path_config((128, 16, "all", 229)), DEV)
    for dt in ("bf16", "f16"):
        with pytest.raises(RuntimeError, match="patch_size 64 or 128"):
            BeatGANsUNetModel(cc.path_config((32, 4, "all", 229), compute_dtype=dt), DEV)
