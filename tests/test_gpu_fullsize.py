"""-m gpu: BASELINE.json's FULL sizes (configs[1]: b=32 x 64-px patches, 50-step DDPM; the test_brn tile:
25 z-chunks x 5x5 patches, P=4), checked through size-independent properties because the CPU oracle
needs ~25 min for one such run:
  * batch independence -- an image's result inside the batch of 32 is bit-identical to the same image run alone
    (the per-launch tile variants differ with the batch, the K-order of every accumulation does not);
  * the oracle on a SUBSET (first 2 images, 3 steps of the 50-step trajectory);
  * the final state of the full reverse loop is the clamped x0 prediction (|x| <= 1), finite, reproducible;
  * locality of the collage decoder (SURVEY.md 8e "global equivalence"): an interior patch of a P=4 tile call
    equals a P=1 call on its own 128x128 state window and 8x8 gene cells;
  * bf16 against fp32 at full size."""
import pytest
import torch

import util
from oracle import teramind_cpu as tc
from teramind_amd import synth
from teramind_amd.config import PathConfig
from teramind_amd.diffusion import SpacedDiffusionBeatGans
from teramind_amd.unet import BeatGANsUNetModel

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
B, T = 32, 50
_M = {}


def model(dtype="f32"):
    if dtype not in _M:
        cfg = PathConfig(compute_dtype=dtype)
        _M[dtype] = BeatGANsUNetModel(cfg, DEV).load_state_dict(util.state_dict(cfg))
    return _M[dtype]


def _inputs(b, P, tag):
    ne = b * (P + 1) ** 2
    x_T = synth.normal(f"{tag}/xT", (b, 4, 64 * P, 64 * P), 11)
    rna = synth.gene_counts(f"{tag}/rna", (ne, 4, 4, 2000), 11)
    return x_T, rna


def _noise(k, b, P=1):
    return synth.normal(f"full/nz{k}", (b * (P + 1) ** 2, 4, 64, 64), 12)


def _run(x_T, rna, steps=None, dtype="f32"):
    smp = SpacedDiffusionBeatGans(T, "ddpm")
    b = x_T.shape[0]
    outs = []
    it = smp.sample_progressive(model(dtype), tuple(x_T.shape), torch.empty(1, 1, 64), rna.to(DEV), None, None,
                                x_T=x_T, step_noise=lambda k: _noise(k, B)[: b * 4])
    for k, img in enumerate(it):
        outs.append(img)
        if steps is not None and k + 1 == steps:
            break
    return outs


def test_config1_batch_independence_and_oracle_subset():
    x_T, rna = _inputs(B, 1, "full")
    full = _run(x_T, rna, steps=3)
    # image i alone: same x_T, its own 4 padded patches of genes, the same per-patch noise
    for i in (0, 17, 31):
        smp = SpacedDiffusionBeatGans(T, "ddpm")
        it = smp.sample_progressive(model(), (1, 4, 64, 64), torch.empty(1, 1, 64), rna[4 * i:4 * i + 4].to(DEV), None, None,
                                    x_T=x_T[i:i + 1], step_noise=lambda k, i=i: _noise(k, B)[4 * i:4 * i + 4])
        for k, img in zip(range(3), it):
            assert torch.equal(img[0], full[k][i]), f"image {i} step {k}: batch of 32 differs from the single run"
    # oracle on the first 2 images, 3 steps
    sd, oc = util.state_dict(PathConfig()), tc.oracle_config_from(PathConfig())
    sch = tc.make_schedule(T, "ddpm")
    x = x_T[:2].clone()
    with torch.inference_mode():
        for k in range(3):
            i = T - 1 - k
            xp = tc.patchify(torch.nn.functional.pad(x, (32, 32, 32, 32)), 64)
            t = torch.full((2,), sch.timestep_map[i], dtype=torch.long)
            pred, _ = tc.unet_forward(sd, oc, xp, t, rna[:8], 2, 2)
            x = tc.sampler_step(sch, "ddpm", xp, pred, i, 1, 1, _noise(k, B)[:8])
            d = (full[k][:2].cpu() - x).abs().max().item()
            assert d < 2e-4, f"step {k}: max |hip - oracle| = {d}"


def test_config1_full_50_step_loop_properties():
    x_T, rna = _inputs(B, 1, "full")
    a = _run(x_T, rna)
    assert len(a) == T
    last = a[-1]
    assert last.shape == (B, 4, 64, 64) and torch.isfinite(last).all()
    # step i=0: posterior_mean_coef1 = 1, coef2 = 0 and no noise -> the clamped x0 prediction (base.py:423-427,476)
    assert last.abs().max().item() <= 1.0 + 1e-6
    assert last.std().item() > 1e-3                       # not collapsed
    b = _run(x_T, rna)
    assert torch.equal(b[-1], last)                        # no atomics / races on the fp32 path: bit reproducible
    assert not torch.equal(a[10], a[11])


def _tile_inputs(b=25, P=4):
    p = P + 1
    x = synth.normal("tile/x", (b * p * p, 4, 64, 64), 5)
    rna = synth.gene_counts("tile/rna", (b * p * p, 4, 4, 2000), 5)
    t = torch.full((b,), 601, dtype=torch.long)
    return x, rna, t


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
def test_tile_config_interior_patch_locality(dtype):
    """mode B geometry at full tile size: 25 z-chunks x 5x5 padded patches (625 encoder, 400 decoder patches) -- the
    shape of BASELINE configs[2] / configs[3] (test_brn.py:188-208), in fp32 and in both 16-bit arithmetic types (the
    whole-brain config runs bf16).  The per-voxel arithmetic does not depend on which other patches share the launch, so
    an interior patch of the P = 4 call equals the P = 1 call on its own four encoder patches."""
    b, P = 25, 4
    p = P + 1
    x, rna, t = _tile_inputs(b, P)
    m = model(dtype)
    big = m(x=x.to(DEV), t=t.to(DEV), rna=rna.to(DEV), imgs=torch.empty((b, 4, 64 * P, 64 * P), device="meta"), patch_size=64).pred
    assert big.shape == (b * P * P, 4, 64, 64) and torch.isfinite(big).all()
    for img, i, j in ((0, 0, 0), (7, 1, 2), (24, 3, 3)):
        idx = [img * p * p + (i + di) * p + (j + dj) for di in (0, 1) for dj in (0, 1)]      # its 4 encoder patches
        small = m(x=x[idx].to(DEV), t=t[:1].to(DEV), rna=rna[idx].to(DEV), imgs=torch.empty((1, 4, 64, 64), device="meta"),
                  patch_size=64).pred
        got = big[img * P * P + i * P + j]
        d = (got - small[0]).abs().max().item()
        assert d <= 1e-5, f"{dtype}: interior patch ({img},{i},{j}): P=4 vs P=1 differ by {d}"


# measured on one MI355X (relative L2 vs the fp32 CPU oracle on eps of std ~0.57): bf16 4e-3, f16 5e-4
TILE_TOL = {"f32": (1e-5, 2e-4), "bf16": (6e-3, 0.04), "f16": (8e-4, 5e-3)}


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
def test_tile_config_oracle_on_two_z_chunks(dtype):
    """The full test_brn tile call (b = 25 z-chunks, P = 4) against the CPU oracle on its first two z-chunks (the oracle
    needs ~7 s per z-chunk at this shape; z-chunks are independent images of the call)."""
    b, P, nz = 25, 4, 2
    p = P + 1
    x, rna, t = _tile_inputs(b, P)
    m = model(dtype)
    big = m(x=x.to(DEV), t=t.to(DEV), rna=rna.to(DEV), imgs=torch.empty((b, 4, 64 * P, 64 * P), device="meta"), patch_size=64).pred
    sd, oc = util.state_dict(PathConfig()), tc.oracle_config_from(PathConfig())
    with torch.inference_mode():
        ref, _ = tc.unet_forward(sd, oc, x[:nz * p * p], t[:nz], rna[:nz * p * p], p, p)
    got = big[:nz * P * P].cpu()
    rel = ((got - ref).norm() / ref.norm()).item()
    mx = (got - ref).abs().max().item()
    print(util.report(f"tile {dtype}", got, ref), "rel_l2=%.3e" % rel)
    assert rel < TILE_TOL[dtype][0] and mx < TILE_TOL[dtype][1], (dtype, rel, mx)
    if dtype != "f32":
        _M.pop(dtype, None)


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
def test_forward_does_not_read_uninitialised_workspace(dtype):
    """The caller-provided workspace is never assumed to be zero: a forward on a workspace poisoned with 0xFF bytes
    (fp32 / bf16 / fp16 NaN patterns) returns the bits of a forward on a zeroed workspace.  (Every channel-pad slot,
    halo cell and pair-padding block a kernel reads is written by its producer first.)"""
    m = model(dtype)
    for b, p1, p2 in ((2, 2, 2), (1, 3, 4)):
        ne = b * p1 * p2
        x = synth.normal(f"poison/x{b}{p1}", (ne, 4, 64, 64), 9).to(DEV)
        rna = synth.gene_counts(f"poison/r{b}{p1}", (ne, 4, 4, 2000), 9).to(DEV)
        t = torch.tensor([77 + 400 * i for i in range(b)], dtype=torch.long, device=DEV)
        kw = dict(x=x, t=t, rna=rna, imgs=torch.empty((b, 4, 64 * (p1 - 1), 64 * (p2 - 1)), device="meta"), patch_size=64,
                  want_pred2=True)
        m(**kw)                                            # sizes the workspace
        [w.fill_(0xFF) for w in m._ws.values()]
        a = m(**kw)
        [w.zero_() for w in m._ws.values()]
        c = m(**kw)
        assert torch.isfinite(a.pred).all() and torch.isfinite(a.pred2).all(), dtype
        assert torch.equal(a.pred, c.pred) and torch.equal(a.pred2, c.pred2), dtype
    if dtype != "f32":
        _M.pop(dtype, None)


@pytest.mark.parametrize("dtype,tol", [("bf16", 6e-3), ("f16", 8e-4)])
def test_bf16_against_fp32_at_full_size(dtype, tol):
    x_T, rna = _inputs(B, 1, "full")
    f = _run(x_T, rna, steps=1)[0]
    h = _run(x_T, rna, steps=1, dtype=dtype)[0]
    rel = ((h - f).pow(2).mean().sqrt() / f.pow(2).mean().sqrt()).item()
    assert rel < tol, rel
    _M.pop(dtype, None)


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
def test_ragged_batches_and_grids_are_per_image_independent(dtype):
    """Odd batch sizes and non-square patch grids (ragged last workgroups in every kernel: patch counts that are not
    multiples of the patches-per-workgroup, voxel counts that are not multiples of the tile): every image's result
    inside the batch is bit-identical to the image run alone, in both arithmetic types."""
    m = model(dtype)
    for b, p1, p2 in ((3, 2, 5), (5, 2, 2), (2, 4, 3), (7, 3, 2)):
        ne_img, nd_img = p1 * p2, (p1 - 1) * (p2 - 1)
        x = synth.normal(f"rag/x{b}{p1}{p2}", (b * ne_img, 4, 64, 64), 3).to(DEV)
        rna = synth.gene_counts(f"rag/r{b}{p1}{p2}", (b * ne_img, 4, 4, 2000), 3).to(DEV)
        t = torch.tensor([(173 * (i + 1)) % 1000 for i in range(b)], dtype=torch.long, device=DEV)
        big = m(x=x, t=t, rna=rna, imgs=torch.empty((b, 4, 64 * (p1 - 1), 64 * (p2 - 1)), device="meta"), patch_size=64,
                want_pred2=True)
        assert torch.isfinite(big.pred).all() and big.pred.shape[0] == b * nd_img
        for i in (0, b - 1):
            one = m(x=x[i * ne_img:(i + 1) * ne_img], t=t[i:i + 1], rna=rna[i * ne_img:(i + 1) * ne_img],
                    imgs=torch.empty((1, 4, 64 * (p1 - 1), 64 * (p2 - 1)), device="meta"), patch_size=64, want_pred2=True)
            assert torch.equal(one.pred, big.pred[i * nd_img:(i + 1) * nd_img]), (dtype, b, p1, p2, i)
            assert torch.equal(one.pred2, big.pred2[i * ne_img:(i + 1) * ne_img]), (dtype, b, p1, p2, i)
    if dtype != "f32":
        _M.pop(dtype, None)
