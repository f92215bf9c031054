"""-m gpu: the HIP UNet / sampler / gene-attention path vs the CPU oracle (oracle/teramind_cpu.py)
on identical hashed weights and seeded synthetic inputs, called through the C-ABI via the
reference-shaped Python adapter.

Tolerance (fp32): the oracle itself sits 1.5e-6 (max abs, outputs O(1)) from the reference
on CPU -- pure fp32 re-association noise through ~50 conv layers.  The MFMA kernels use a
different but equally valid fp32 summation order, so the bound is set at 2e-4 max-abs on
outputs of std ~0.57 (|max| ~2.5), ~100x above the observed drift."""
import os

import numpy as np
import pytest
import torch

import util
from oracle import teramind_cpu as tc
from teramind_amd import synth
from teramind_amd.config import PathConfig
from teramind_amd.unet import BeatGANsUNetModel, GeneAttnModel

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ATOL = 2e-4

_MODEL = {}


def hip_model():
    if "m" not in _MODEL:
        cfg = PathConfig()
        _MODEL["m"] = BeatGANsUNetModel(cfg, DEV).load_state_dict(util.state_dict(cfg))
    return _MODEL["m"]


def make_inputs(b, P, seed=0):
    p = P + 1
    ne = b * p * p
    x = synth.normal("x", (ne, 4, 64, 64), seed)
    rna = synth.gene_counts("rna", (ne, 4, 4, 2000), seed)
    t = torch.tensor([(137 * (i + 1) + 61 * seed) % 1000 for i in range(b)], dtype=torch.long)
    return x, t, rna


def test_unet_forward_taps_b1_P1(out_dir, tmp_path):
    """One encoder-grid of 4 patches -> 1 interior patch; every block output compared."""
    cfg = PathConfig()
    oc = tc.oracle_config_from(cfg)
    sd = util.state_dict(cfg)
    x, t, rna = make_inputs(1, 1)
    taps = {}
    with torch.inference_mode():
        ref, ref2 = tc.unet_forward(sd, oc, x, t, rna, 2, 2, want_pred2=True, taps=taps)
    os.environ["TM_DEBUG_DIR"] = str(tmp_path)
    try:
        out = hip_model()(x=x.to(DEV), t=t.to(DEV), rna=rna.to(DEV), imgs=torch.zeros(1, 4, 64, 64), patch_size=64,
                          want_pred2=True)
        torch.cuda.synchronize()
    finally:
        del os.environ["TM_DEBUG_DIR"]
    lines, worst = [], 0.0
    for name, refv in taps.items():
        f = tmp_path / f"{name}.bin"
        if not f.exists():
            continue
        got = torch.from_numpy(np.fromfile(f, dtype=np.float32)).reshape(refv.shape)
        lines.append(util.report(name, got, refv))
        worst = max(worst, (got - refv).abs().max().item() / max(1.0, refv.abs().max().item()))
    lines.append(util.report("pred", out.pred, ref))
    lines.append(util.report("pred2", out.pred2, ref2))
    with open(os.path.join(out_dir, "unet_taps_b1_P1.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))
    assert torch.allclose(out.pred.cpu(), ref, atol=ATOL, rtol=0), lines[-2]
    assert torch.allclose(out.pred2.cpu(), ref2, atol=ATOL, rtol=0), lines[-1]
    assert worst < 1e-3


@pytest.mark.parametrize("b,P", [(2, 1), (1, 2)])
def test_unet_forward_shapes(b, P):
    cfg = PathConfig()
    oc = tc.oracle_config_from(cfg)
    sd = util.state_dict(cfg)
    x, t, rna = make_inputs(b, P, seed=3)
    with torch.inference_mode():
        ref, _ = tc.unet_forward(sd, oc, x, t, rna, P + 1, P + 1)
    out = hip_model()(x=x.to(DEV), t=t.to(DEV), rna=rna.to(DEV), imgs=torch.zeros(b, 4, 64 * P, 64 * P), patch_size=64)
    assert out.pred2 is None
    assert torch.allclose(out.pred.cpu(), ref, atol=ATOL, rtol=0), util.report("pred", out.pred, ref)


def test_unet_rna_coo_input_equals_dense():
    x, t, rna = make_inputs(1, 1, seed=5)
    m = hip_model()
    kw = dict(x=x.to(DEV), t=t.to(DEV), imgs=torch.zeros(1, 4, 64, 64), patch_size=64)
    a = m(rna=rna.to(DEV), **kw).pred
    b = m(rna=synth.dense_to_coo(rna), **kw).pred
    assert torch.equal(a, b)


def test_gene_attention_maps():
    cfg = PathConfig()
    oc = tc.oracle_config_from(cfg)
    sdv = util.state_dict(cfg, vis_only=True)
    rna = synth.gene_counts("rna_vis", (3, 4, 4, 2000), 1, density=0.05)
    with torch.inference_mode():
        ref_a, ref_r = tc.gene_attention_maps(sdv, oc, rna)
    m = GeneAttnModel(cfg, DEV).load_state_dict(sdv, strict=False)
    a, r = m(rna=rna.to(DEV), imgs=torch.zeros(1, 4, 64, 64))
    assert a.shape == (4, 3, 229, 229)
    assert torch.equal(r.cpu(), ref_r)
    # softmax probabilities ~1/229: absolute 1e-7 is ~2e-5 relative
    assert torch.allclose(a.cpu(), ref_a, atol=1e-7, rtol=1e-4), util.report("attn", a, ref_a)
    assert torch.allclose(a.sum(-1).cpu(), torch.ones(4, 3, 229), atol=1e-5)


def test_arena_view_and_single_rank_process_group():
    """model.arena() is a zero-copy uint8 view of the packed weights (what rank 0 broadcasts over
    RCCL in place of DDP's parameter broadcast, test_brn.py:149); exercised on a 1-rank nccl group."""
    import torch.distributed as dist
    m = hip_model()
    a = m.arena()
    assert a.dtype == torch.uint8 and a.is_cuda and a.numel() > 850e6
    before = a[:4096].clone()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        dist.broadcast(a, src=0)
        dist.barrier()
        t = torch.tensor([1.5], dtype=torch.float64, device=DEV)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t) == 1.5
    finally:
        dist.destroy_process_group()
    assert torch.equal(a[:4096], before)


@pytest.mark.parametrize("dtype,rel_tol,abs_tol", [("bf16", 6e-3, 0.03), ("f16", 8e-4, 4e-3)])
def test_unet_forward_bf16_convs(dtype, rel_tol, abs_tol):
    """TM_DTYPE_BF16 (BASELINE config 4): bf16 weights + bf16 operands of the convs / Linears / attention, fp32
    accumulate / residual stream.  Tolerance is bf16 operand rounding (2^-9 relative per op): measured 3.8e-3
    relative L2 / 1.2e-2 max-abs vs the fp32 oracle on outputs of std 0.57; bounds 6e-3 / 0.03.
    TM_DTYPE_F16 (IEEE half, the reference's own autocast arithmetic): three more mantissa bits, measured 4.8e-4 /
    1.5e-3; bounds 8e-4 / 4e-3."""
    cfg = PathConfig(compute_dtype=dtype)
    oc = tc.oracle_config_from(cfg)
    sd = util.state_dict(cfg)
    m = BeatGANsUNetModel(cfg, DEV).load_state_dict(sd)
    for b, P, seed in [(1, 1, 0), (1, 2, 3)]:
        x, t, rna = make_inputs(b, P, seed)
        with torch.inference_mode():
            ref, _ = tc.unet_forward(sd, oc, x, t, rna, P + 1, P + 1)
        out = m(x=x.to(DEV), t=t.to(DEV), rna=rna.to(DEV), imgs=torch.zeros(b, 4, 64 * P, 64 * P), patch_size=64).pred.cpu()
        rel = ((out - ref).norm() / ref.norm()).item()
        print(util.report(f"{dtype} b{b} P{P}", out, ref), "rel_l2=%.3e" % rel)
        assert rel < rel_tol and (out - ref).abs().max() < abs_tol, util.report(dtype, out, ref)


@pytest.mark.parametrize("stain", ["DAPI", "PolyT"])
def test_unet_single_stain_config(stain):
    """stain != 'all': one stain x 2 z-slices = 2 image channels (config surface of train.py:33-35);
    the oracle was validated against the reference for these configs (1.97e-6)."""
    cfg = PathConfig(stain=stain)
    oc = tc.oracle_config_from(cfg)
    sd = util.state_dict(cfg)
    m = BeatGANsUNetModel(cfg, DEV).load_state_dict(sd)
    b, P = 2, 1
    ne = b * (P + 1) ** 2
    x = synth.normal("x", (ne, cfg.in_channels, 64, 64), 7)
    rna = synth.gene_counts("rna", (ne, 4, 4, 2000), 7)
    t = torch.tensor([11, 950])
    with torch.inference_mode():
        ref, ref2 = tc.unet_forward(sd, oc, x, t, rna, 2, 2, want_pred2=True)
    out = m(x=x.to(DEV), t=t.to(DEV), rna=rna.to(DEV), imgs=torch.zeros(b, cfg.in_channels, 64, 64), patch_size=64, want_pred2=True)
    assert out.pred.shape == (b, 2, 64, 64)
    assert torch.allclose(out.pred.cpu(), ref, atol=ATOL, rtol=0), util.report("pred", out.pred, ref)
    assert torch.allclose(out.pred2.cpu(), ref2, atol=ATOL, rtol=0), util.report("pred2", out.pred2, ref2)


def test_unet_batch_of_images_with_interior_grid_and_distinct_timesteps():
    """b = 2 images of 2x3 interior patches (non-square grid), a different t per image: exercises the
    per-image emb modulation index, the collage index map with p1 != p2 and ragged tile counts."""
    cfg = PathConfig()
    oc = tc.oracle_config_from(cfg)
    sd = util.state_dict(cfg)
    b, p1, p2 = 2, 3, 4
    ne = b * p1 * p2
    x = synth.normal("x23", (ne, 4, 64, 64), 1)
    rna = synth.gene_counts("rna23", (ne, 4, 4, 2000), 1)
    t = torch.tensor([3, 871])
    with torch.inference_mode():
        ref, _ = tc.unet_forward(sd, oc, x, t, rna, p1, p2)
    out = hip_model()(x=x.to(DEV), t=t.to(DEV), rna=rna.to(DEV), imgs=torch.zeros(b, 4, 64 * (p1 - 1), 64 * (p2 - 1)), patch_size=64)
    assert out.pred.shape == (b * 2 * 3, 4, 64, 64)
    assert torch.allclose(out.pred.cpu(), ref, atol=ATOL, rtol=0), util.report("pred", out.pred, ref)


def test_forward_argument_validation():
    m = hip_model()
    x, t, rna = make_inputs(1, 1)
    with pytest.raises(ValueError):
        m(x=x[:3].to(DEV), t=t.to(DEV), rna=rna.to(DEV), imgs=torch.zeros(1, 4, 64, 64), patch_size=64)
    with pytest.raises(ValueError):
        m(x=x.to(DEV), t=t.to(DEV), rna=rna[:, :3].to(DEV), imgs=torch.zeros(1, 4, 64, 64), patch_size=64)
    with pytest.raises(ValueError):
        m(x=x.to(DEV), t=t.to(DEV), rna=rna.to(DEV), imgs=torch.zeros(1, 4, 64, 64), patch_size=32)
    with pytest.raises(NotImplementedError):
        m(x=x.to(DEV), t=t.to(DEV), rna=rna.to(DEV), imgs=torch.zeros(1, 4, 64, 64), patch_size=64, do_train=True)
    with pytest.raises(RuntimeError):
        BeatGANsUNetModel(PathConfig(), DEV)(x=x.to(DEV), t=t.to(DEV), rna=rna.to(DEV), imgs=torch.zeros(1, 4, 64, 64))
    with pytest.raises(RuntimeError):
        BeatGANsUNetModel(PathConfig(), DEV).load_state_dict({"bogus.key": torch.zeros(1)})


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
def test_precomputed_rna_pyramid_is_bit_identical(dtype):
    """tm_rna_pyramid + tm_unet_forward_rna (the RNA conditioning computed once for the steps of a reverse loop) return
    the bits of tm_unet_forward, which recomputes it per call like the reference (unet_ours.py:376); the sampler's mode-A
    loop uses it (diffusion.sample_progressive)."""
    cfg = PathConfig(compute_dtype=dtype)
    m = hip_model() if dtype == "f32" else BeatGANsUNetModel(cfg, DEV).load_state_dict(util.state_dict(cfg))
    b, P = 2, 2
    x, t, rna = make_inputs(b, P, seed=9)
    shp = torch.zeros(b, 4, 64 * P, 64 * P)
    a = m(x=x.to(DEV), t=t.to(DEV), rna=rna.to(DEV), imgs=shp, patch_size=64, want_pred2=True)
    pyr = m.precompute_rna(rna.to(DEV), b, imgs=shp, patch_size=64)
    for tt in (t, torch.tensor([5, 990])):
        ref = m(x=x.to(DEV), t=tt.to(DEV), rna=rna.to(DEV), imgs=shp, patch_size=64, want_pred2=True)
        got = m(x=x.to(DEV), t=tt.to(DEV), rna=pyr, imgs=shp, patch_size=64, want_pred2=True)
        assert torch.equal(got.pred, ref.pred) and torch.equal(got.pred2, ref.pred2)
    assert torch.equal(a.pred, m(x=x.to(DEV), t=t.to(DEV), rna=pyr, imgs=shp, patch_size=64).pred)
    with pytest.raises(ValueError):
        m(x=x[:9].to(DEV), t=t[:1].to(DEV), rna=pyr, imgs=torch.zeros(1, 4, 128, 128), patch_size=64)


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
def test_cached_rna_level0_is_bit_identical(dtype):
    """tm_rna_level0 + tm_unet_forward_level0 (gene attention -> down_z -> Upsample kept per tile across the steps of a sweep,
    TileSweep(cache_level0=True)) return the bits of tm_unet_forward; a level 0 of another (b, p1, p2) is refused."""
    cfg = PathConfig(compute_dtype=dtype)
    m = hip_model() if dtype == "f32" else BeatGANsUNetModel(cfg, DEV).load_state_dict(util.state_dict(cfg))
    b, P = 2, 2
    x, t, rna = make_inputs(b, P, seed=13)
    shp = torch.zeros(b, 4, 64 * P, 64 * P)
    l0 = m.precompute_rna_level0(rna.to(DEV), b, imgs=shp, patch_size=64)
    nbytes = b * (P + 1) ** 2 * 29 * 2 * 8 * 8 * 8 * (4 if dtype == "f32" else 2)        # [Ne][29 blocks of 8][Z 2][8][8][8]
    assert l0.buf.numel() == nbytes
    for tt in (t, torch.tensor([7, 950])):
        ref = m(x=x.to(DEV), t=tt.to(DEV), rna=rna.to(DEV), imgs=shp, patch_size=64, want_pred2=True)
        got = m(x=x.to(DEV), t=tt.to(DEV), rna=l0, imgs=shp, patch_size=64, want_pred2=True)
        assert torch.equal(got.pred, ref.pred) and torch.equal(got.pred2, ref.pred2)
    with pytest.raises(ValueError):
        m(x=x[:9].to(DEV), t=t[:1].to(DEV), rna=l0, imgs=torch.zeros(1, 4, 128, 128), patch_size=64)


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
def test_overlap_streams_is_bit_identical(dtype):
    """model.overlap_streams = 2: the two halves of a call's images on two HIP streams (the call's workspace cut in two) return the bits of the
    one-stream call -- dense genes and cached level 0, odd image count, pred2 too; a call on a caller-chosen side stream works and
    the stream's own workspace is kept apart from the default stream's."""
    cfg = PathConfig(compute_dtype=dtype)
    m = BeatGANsUNetModel(cfg, DEV).load_state_dict(util.state_dict(cfg))
    b, P = 3, 2
    x, t, rna = make_inputs(b, P, seed=17)
    x, rna, t = x.to(DEV), rna.to(DEV), torch.tensor([7, 950, 333]).to(DEV)
    shp = torch.zeros(b, 4, 64 * P, 64 * P)
    l0 = m.precompute_rna_level0(rna, b, imgs=shp, patch_size=64)
    ref = m(x=x, t=t, rna=rna, imgs=shp, patch_size=64, want_pred2=True)
    m.overlap_streams = 2
    for r in (rna, l0):
        got = m(x=x, t=t, rna=r, imgs=shp, patch_size=64, want_pred2=True)
        torch.cuda.synchronize()
        assert torch.equal(got.pred, ref.pred) and torch.equal(got.pred2, ref.pred2)
    assert len(m._ws) == 1                                  # both halves live in the caller's stream's workspace
    m.overlap_streams = 1
    s = torch.cuda.Stream(device=DEV)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        got = m(x=x, t=t, rna=rna, imgs=shp, patch_size=64)
    s.synchronize()
    assert torch.equal(got.pred, ref.pred) and len(m._ws) == 2


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
def test_half_resolution_conditioning_is_bit_identical(dtype, tmp_path):
    """The AttnBlock computes SiLU(cond), the 7C adaLN modulation, the cross-cond chunk, k and v once per aligned
    2 x 2 voxel block (cond is a nearest-x2 upsampled RNA level: unet_ours.py:290-295, MBAblocks.py:463-466,472-479,487) and
    reads them at (z, y >> 1, x >> 1).  TM_ATTN_HALF=0 switches the full-resolution form back on (read once per process,
    hence the child): both must agree bit for bit, through the plain encoder blocks and the collage decoder."""
    import subprocess
    import sys
    import half_worker
    here = os.path.dirname(os.path.abspath(__file__))
    out = str(tmp_path / "full.pt")
    env = dict(os.environ, TM_ATTN_HALF="0")
    r = subprocess.run([sys.executable, os.path.join(here, "half_worker.py"), dtype, out], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    full = torch.load(out, weights_only=True)
    assert os.environ.get("TM_ATTN_HALF") in (None, "1")
    pred, pred2 = half_worker.run(dtype)
    assert torch.equal(pred, full["pred"]) and torch.equal(pred2, full["pred2"])
