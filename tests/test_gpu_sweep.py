"""-m gpu: the tiled sweep (test_brn data path re-designed with resident state) on the GPU with
the HIP model, against the same sweep driven by the CPU oracle."""
import pytest
import torch

import util
from oracle import teramind_cpu as tc
from teramind_amd.brain import TileSweep, synthetic_gene_provider
from teramind_amd.config import PathConfig
from teramind_amd.diffusion import SpacedDiffusionBeatGans
from teramind_amd.unet import BeatGANsUNetModel

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T, SLC = 2, 4


class OracleSampler:
    """sampler.sample(...)-shaped wrapper of the CPU oracle (mode B)."""

    def __init__(self, cfg, sd):
        self.oc, self.sd, self.sch = tc.oracle_config_from(cfg), sd, tc.make_schedule(T, "ddim")

    def sample(self, model=None, shape=None, imgs=None, noise=None, r_start=None, patch_size=64, idx=None, **kw):
        n, c, H, W = shape
        P1, P2 = H // patch_size, W // patch_size
        t = torch.full((n,), self.sch.timestep_map[idx], dtype=torch.long)
        with torch.inference_mode():
            pred, _ = tc.unet_forward(self.sd, self.oc, imgs, t, r_start, P1 + 1, P2 + 1)
            return tc.sampler_step(self.sch, "ddim", imgs, pred, idx, P1, P2)


def test_tile_sweep_hip_vs_oracle():
    cfg = PathConfig()
    sd = util.state_dict(cfg)
    genes = synthetic_gene_provider(cfg, total_slc=SLC)
    kw = dict(hst=256, wst=512, hnm=1, wnm=2, total_epochs=T, total_slc=SLC)
    ref = TileSweep(cfg, OracleSampler(cfg, sd), None, genes, device="cpu", **kw).test()
    model = BeatGANsUNetModel(cfg, DEV).load_state_dict(sd)
    got = TileSweep(cfg, SpacedDiffusionBeatGans(T, "ddim"), model, genes, device=DEV, batch_tiles=2, **kw).test()
    assert got.shape == (SLC * 2, 256, 512)
    # two DDIM steps from t=500: forward noise x sqrt(1/abar-1) stays O(1e-4); state is fp16-rounded
    # after every step so a 1-ulp fp16 flip (1e-3 at |x|~1) is the granularity
    d = (got.cpu() - ref).abs()
    assert d.max() <= 2e-3 and d.mean() <= 2e-5, util.report("sweep", got, ref)


# (max, mean) bounds on |state - oracle sweep| after the two DDIM steps; the state is fp16-rounded after every step, x0 is
# clamped to [-1, 1] and eps errors enter x0 with a factor sqrt(1/abar - 1) ~ 3.4 at t = 500 (16-bit eps errors: bf16
# ~2e-3 absolute, f16 ~3e-4)
SWEEP16_TOL = {"bf16": (0.15, 6e-3), "f16": (0.03, 1e-3)}


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_tile_sweep_16bit_whole_brain_layout_vs_oracle(dtype):
    """BASELINE configs[3] in miniature: the tile sweep in the whole-brain configuration -- 16-bit UNet arithmetic
    (bf16 as the config names it, f16 as the reference's autocast runs it) and ONE float16 state canvas -- against the
    sweep driven by the fp32 CPU oracle (test_brn.py:174-226 per tile, :232-255 per step)."""
    cfg = PathConfig(compute_dtype=dtype)
    sd = util.state_dict(cfg)
    genes = synthetic_gene_provider(cfg, total_slc=SLC)
    kw = dict(hst=256, wst=512, hnm=2, wnm=1, total_epochs=T, total_slc=SLC)
    ref = TileSweep(cfg, OracleSampler(cfg, sd), None, genes, device="cpu", **kw).test()
    model = BeatGANsUNetModel(cfg, DEV).load_state_dict(sd)
    sw = TileSweep(cfg, SpacedDiffusionBeatGans(T, "ddim"), model, genes, device=DEV, batch_tiles=1, state="fp16", **kw)
    got = sw.test()
    assert got.dtype == torch.float16 and got.shape == (SLC * 2, 512, 256) and sw.nxt is None
    d = (got.float().cpu() - ref).abs()
    print(util.report(f"sweep {dtype}", got, ref))
    assert d.max() <= SWEEP16_TOL[dtype][0] and d.mean() <= SWEEP16_TOL[dtype][1], util.report("sweep " + dtype, got, ref)
    assert float(got.float().abs().max()) <= 1.0 + 1e-3          # last DDIM step: the clamped x0 prediction


def test_tile_sweep_single_fp16_canvas_equals_two_canvas_state():
    """state='fp16' (the whole-brain memory layout: one float16 canvas, rows committed one row late) on the GPU
    with the HIP model: bit-identical to the default two-canvas fp32 state."""
    cfg = PathConfig()
    sd = util.state_dict(cfg)
    genes = synthetic_gene_provider(cfg, total_slc=SLC)
    model = BeatGANsUNetModel(cfg, DEV).load_state_dict(sd)
    kw = dict(hst=256, wst=512, hnm=2, wnm=2, total_epochs=T, total_slc=SLC, device=DEV, init="device")
    a = TileSweep(cfg, SpacedDiffusionBeatGans(T, "ddim"), model, genes, batch_tiles=1, **kw).test()
    sw = TileSweep(cfg, SpacedDiffusionBeatGans(T, "ddim"), model, genes, batch_tiles=3, state="fp16", **kw)
    b = sw.test()
    assert b.dtype == torch.float16 and sw.nxt is None
    assert torch.equal(b.float(), a)


@pytest.mark.parametrize("rna_slc,state_slc,gene_slc", [(1, 2, 2), (8, 8, 10)])
def test_tile_sweep_other_rna_slc(rna_slc, state_slc, gene_slc):
    """The sweep for the per-slice model (rna_slc 1) and a z_size-4 model (rna_slc 8: state slices = gene slices - 2,
    like the reference's 48 / 50): one tile, one DDIM step, HIP vs the oracle-driven sweep."""
    cfg = PathConfig(rna_slc=rna_slc)
    sd = util.state_dict(cfg)
    genes = synthetic_gene_provider(cfg, total_slc=gene_slc)
    kw = dict(hst=256, wst=512, hnm=1, wnm=1, total_epochs=T, total_slc=state_slc)

    class OneStep(OracleSampler):
        pass

    ref_sw = TileSweep(cfg, OneStep(cfg, sd), None, genes, device="cpu", **kw)
    ref_sw.step()
    ref = ref_sw.local_state()
    model = BeatGANsUNetModel(cfg, DEV).load_state_dict(sd)
    sw = TileSweep(cfg, SpacedDiffusionBeatGans(T, "ddim"), model, genes, device=DEV, **kw)
    sw.step()
    got = sw.local_state()
    assert got.shape == (state_slc * 2, 256, 256)
    d = (got.cpu() - ref).abs()
    assert d.max() <= 2e-3 and d.mean() <= 2e-5, util.report("sweep", got, ref)


@pytest.mark.parametrize("dtype,state", [("f32", "fp32x2"), ("bf16", "fp16"), ("f16", "fp16")])
def test_share_halo_window_is_bit_identical_to_per_tile_calls(dtype, state):
    """TileSweep(share_halo=True): the tiles of a call as ONE window (the patch columns two neighbouring tiles share --
    right halo of one, left interior edge of the next, utils/MBADataset_tst.py:91-123 -- go through the encoder once)
    == one model call per tile as test_brn.py:174-226 does it, bit for bit, on gene tiles that agree where they overlap."""
    from teramind_amd.brain import consistent_gene_provider
    cfg = PathConfig(compute_dtype=dtype)
    model = BeatGANsUNetModel(cfg, DEV).load_state_dict(util.state_dict(cfg))
    genes = consistent_gene_provider(cfg, DEV, total_slc=SLC, density=0.05)
    # neighbouring tiles see the same cells where their gene grids overlap
    a, b = genes(3, 5), genes(3, 6)
    assert torch.equal(a[:, 16:20], b[:, 0:4]) and float(a.sum()) > 0
    kw = dict(hst=512, wst=1024, hnm=2, wnm=3, total_epochs=T, total_slc=SLC, device=DEV, init="device", state=state)
    per_tile = TileSweep(cfg, SpacedDiffusionBeatGans(T, "ddim"), model, genes, batch_tiles=1, **kw).test()
    window = TileSweep(cfg, SpacedDiffusionBeatGans(T, "ddim"), model, genes, batch_tiles=3, share_halo=True, **kw).test()
    ragged = TileSweep(cfg, SpacedDiffusionBeatGans(T, "ddim"), model, genes, batch_tiles=2, share_halo=True, **kw).test()
    assert torch.equal(per_tile, window) and torch.equal(per_tile, ragged)
    # windows of several tile rows: the halo rows between them are shared too
    kw3 = dict(kw, hnm=3)
    per_tile3 = TileSweep(cfg, SpacedDiffusionBeatGans(T, "ddim"), model, genes, batch_tiles=1, **kw3).test()
    rows2 = TileSweep(cfg, SpacedDiffusionBeatGans(T, "ddim"), model, genes, batch_tiles=2, batch_rows=2, share_halo=True, **kw3).test()
    rows3 = TileSweep(cfg, SpacedDiffusionBeatGans(T, "ddim"), model, genes, batch_tiles=3, batch_rows=3, share_halo=True, **kw3).test()
    assert torch.equal(per_tile3, rows2) and torch.equal(per_tile3, rows3)
    # z_group: a window's images (z-chunks) through the model one at a time -- the workspace of one image, the same bits
    zg = TileSweep(cfg, SpacedDiffusionBeatGans(T, "ddim"), model, genes, batch_tiles=3, batch_rows=3, share_halo=True, z_group=1,
                   cache_level0=True, **kw3).test()
    zg_stacked = TileSweep(cfg, SpacedDiffusionBeatGans(T, "ddim"), model, genes, batch_tiles=2, z_group=3, **kw3).test()
    assert torch.equal(per_tile3, zg) and torch.equal(per_tile3, zg_stacked)


@pytest.mark.parametrize("dtype,state", [("f32", "fp32x2"), ("bf16", "fp16")])
def test_cached_level0_sweep_is_bit_identical(dtype, state):
    """TileSweep(cache_level0=True): level 0 of the RNA conditioning (gene attention -> down_z -> Upsample,
    unet_ours.py:298-310) of every model call is computed in the first step and reused in the following ones (the genes of
    a tile do not change between steps, test_brn.py:232-255) -- the same states, bit for bit, for stacked tiles and for
    shared-halo windows."""
    from teramind_amd.brain import consistent_gene_provider
    cfg = PathConfig(compute_dtype=dtype)
    model = BeatGANsUNetModel(cfg, DEV).load_state_dict(util.state_dict(cfg))
    genes = consistent_gene_provider(cfg, DEV, total_slc=SLC, density=0.05)
    T3 = 3
    kw = dict(hst=512, wst=1024, hnm=2, wnm=2, total_epochs=T3, total_slc=SLC, device=DEV, init="device", state=state)
    mk = lambda **k: TileSweep(cfg, SpacedDiffusionBeatGans(T3, "ddim"), model, genes, **kw, **k)
    plain = mk(batch_tiles=2).test()
    sw = mk(batch_tiles=2, cache_level0=True)
    cached = sw.test()
    assert len(sw._level0) == 2 and torch.equal(plain, cached)
    win = mk(batch_tiles=2, batch_rows=2, share_halo=True, cache_level0=True).test()
    assert torch.equal(plain, win)
