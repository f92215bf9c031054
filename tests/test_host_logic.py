"""CPU: host logic of the package (teramind_amd.diffusion / tiles / weights / brain) against the
reference-minted fixtures and the oracle.  Integer paths are compared bit for bit."""
import json
import os

import numpy as np
import pytest
import torch

import util
from oracle import teramind_cpu as tc
from teramind_amd import diffusion, tiles
from teramind_amd.config import PathConfig, parse_ckpt_dir_name, prep_config_parm
from teramind_amd.weights import hashed_tensor, param_spec, strip_lightning_state_dict

G = util.GOLDEN
INT = json.load(open(os.path.join(G, "integer_paths.json")))


def test_param_spec_matches_reference_state_dict():
    gold = json.load(open(os.path.join(G, "state_dict_keys_638850_64_229_all_4_ours.json")))
    assert [(k, tuple(s)) for k, s in gold] == param_spec(PathConfig())
    assert len(param_spec(PathConfig(), vis_only=True)) == 18


def test_config_surface():
    c = prep_config_parm("", 1, 64, 8, "all", "638850", 229, 4)
    assert (c.z_size, c.in_channels, c.gn_sz, c.gene_hidden, c.down_z_kernel) == (2, 4, 4, 64, 3)
    assert c.name == "638850_64_229_all_4_ours"
    assert parse_ckpt_dir_name(c.name) == c
    with pytest.raises(NotImplementedError):
        PathConfig(patch_size=48)
    assert PathConfig(stain="DAPI", rna_slc=1).in_channels == 1


def test_space_timesteps_and_tables_bit_exact():
    for key, rec in INT["space_timesteps"].items():
        assert sorted(diffusion.space_timesteps(rec["T"], rec["section_counts"])) == rec["steps"], key
    with pytest.raises(ValueError):
        diffusion.space_timesteps(1000, "ddim999")
    tabs = np.load(os.path.join(G, "tables.npz"))
    for T, gen in [(15, "ddim"), (50, "ddim"), (50, "ddpm"), (1000, "ddpm")]:
        s = diffusion.SpacedDiffusionBeatGans(T, gen)
        if f"{gen}{T}" in INT["timestep_map"]:
            assert s.timestep_map == INT["timestep_map"][f"{gen}{T}"]
        for name in ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
                     "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
                     "posterior_mean_coef1", "posterior_mean_coef2"]:
            assert np.array_equal(getattr(s, name), tabs[f"{gen}{T}/{name}"]), (gen, T, name)
        sch = tc.make_schedule(T, gen)
        assert np.array_equal(s.model_log_variance, sch.model_log_variance)


def test_sparse_repatch_bit_exact_and_non_mutating():
    r = INT["sparse_repatch"]
    crd = torch.tensor(r["crd_in"])
    keep = crd.clone()
    _, out, ssz = diffusion.sparse_repatch((torch.zeros(crd.shape[1]), crd, torch.Size(r["ssz"])), r["sz"])
    assert out.tolist() == r["crd_out"] and list(ssz) == r["ssz_out"]
    assert torch.equal(crd, keep)


def test_lcg_seeds_names_partition():
    for k, v in INT["lcg"].items():
        assert tiles.lcg(int(k)) == v
    assert [n + ".npz" for n in tiles.gene_tile_names(hst=256, wst=512, hnm=2, wnm=3)] == INT["gn_sublst"]
    assert tiles.state_tile_name(1, 2) == "256_512_512_768"
    assert tiles.row_block_partition(286, 8) == [(0, 36), (36, 72), (72, 108), (108, 144), (144, 180), (180, 216), (216, 251), (251, 286)]
    assert tiles.row_block_partition(32, 8)[3] == (12, 16)
    assert tiles.row_block_partition(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]


def _digest_matches(t, d, atol=0.0):
    f = t.detach().float().reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, len(d["samples"])).long()
    assert list(t.shape) == d["shape"]
    assert torch.allclose(f[idx], torch.tensor(d["samples"]), atol=atol, rtol=0)
    assert abs(float(f.double().mean()) - d["mean"]) <= 1e-9 + atol


def test_run_batch_layout_maps_vs_reference():
    """z-chunk / patchify / regroup index maps (a23) against what the reference's own
    Tester._run_batch handed to its sampler and wrote out (captured by make_golden.py)."""
    rb = INT["run_batch"]
    b = 1
    tile = torch.arange(b * 320 * 320 * 100, dtype=torch.float32).reshape(b, 320, 320, 100)
    gen = torch.Generator().manual_seed(9)
    ssz = (b, 20, 20, 26000)
    crd = torch.stack([torch.randint(0, ssz[k], (4000,), generator=gen) for k in range(4)])
    crd = torch.unique(crd, dim=1)
    dat = (crd[1] * 20 * 26000 + crd[2] * 26000 + crd[3] + 1).float()
    rna = torch.sparse_coo_tensor(crd, dat, ssz).to_dense()
    x, r, shape = tiles.run_batch_inputs(tile, rna, 64, 4, 50, 4)
    assert list(shape) == rb["shape"] and list(x.shape) == rb["imgs_shape"] and list(r.shape) == rb["rna_shape"]
    assert x[:, 1, 5, 7].long().tolist() == rb["imgs_probe"]
    _digest_matches(x, rb["imgs_digest"])
    assert float(x.double().sum()) == rb["imgs_sum"]
    assert int((r != 0).sum()) == rb["rna_nonzero"] and float(r.double().sum()) == rb["rna_sum"]
    for n, probe in zip((0, 7, 24 * 25 + 12, 624), rb["rna_probe"]):
        got = [int(v) for v in r[n].nonzero()[:3].reshape(-1).tolist()] + [float(r[n].max())]
        assert got == probe
    # oracle restatement agrees too
    assert torch.equal(tc.zchunk_state(tile, 50, 4), tiles.zchunk_state(tile, 50, 4))
    assert torch.equal(tc.zchunk_rna(rna, 4), tiles.zchunk_rna(rna, 4))
    # output regroup + fp16 cast
    n = shape[0]
    out = (torch.arange(n * 4 * 256 * 256, dtype=torch.float32).reshape(n, 4, 256, 256) % 2039) / 64.0
    saved = tiles.regroup_output(out, b, 2).half()
    assert list(saved.shape[1:]) == rb["saved_shape"] and rb["saved_dtype"] == "float16"
    _digest_matches(saved[0].float(), rb["saved_digest"])
    assert torch.equal(tc.unchunk_state(out, b, 2), tiles.regroup_output(out, b, 2))


def test_initial_noise_and_halo_assembly_vs_reference():
    """Step-0 padded tiles (LCG-seeded mt19937 noise of the 3x3 neighbourhood, -1 outside the ROI)
    out of the resident canvas == MBADataset_tst._pad_im(roi, 0) of the reference."""
    from teramind_amd.brain import TileSweep
    rec = INT["pad_im_step0"]
    sw = TileSweep(PathConfig(), sampler=None, model=None, gene_provider=None, hst=rec["hst"] * 256, wst=rec["wst"] * 256,
                   hnm=rec["hnm"], wnm=rec["wnm"], total_epochs=15)
    for name, d in rec["tiles"].items():
        r0, _, c0, _ = map(int, name.split("_"))
        lr, c = r0 // 256 - rec["hst"], c0 // 256 - rec["wst"]
        win = sw._window(lr, c)
        assert int((win == -1).sum()) == d["n_minus1"]
        _digest_matches(win, {k: d[k] for k in ("shape", "mean", "samples")}, atol=0.0)


def test_hashed_generator_is_pure_function_of_key_and_index():
    a = hashed_tensor("input_blocks.1.0.in_layers.2.weight", (64, 96, 3, 3, 3), 0)
    from teramind_amd.weights import hashed_uniform
    u = hashed_uniform("input_blocks.1.0.in_layers.2.weight", 64 * 96 * 27, 0)
    assert np.array_equal(u[:1000], hashed_uniform("input_blocks.1.0.in_layers.2.weight", 1000, 0))
    assert np.allclose(a.reshape(-1), u * np.sqrt(3.0) / (96 * 27) ** 0.5, rtol=1e-6)
    assert abs(float(a.std()) - 1 / (96 * 27) ** 0.5) < 2e-4
    z = hashed_tensor("input_blocks.1.0.out_layers.3.weight", (64, 64, 3, 3, 3), 0)
    assert float(np.abs(z).max()) > 0            # zero_module convs are overwritten
    sd = strip_lightning_state_dict({"state_dict": {"model.out.0.weight": 1, "ema_model.out.0.weight": 2}})
    assert sd == {"out.0.weight": 1}


@pytest.mark.parametrize("z", [1, 8, 16])
def test_run_batch_layout_maps_other_z_vs_reference(z):
    """rna_slc 1 (per-slice path) and 8 / 16 (48 state slices, 50 gene slices + z padding): the index maps against
    what the reference's Tester._run_batch produced (oracle/make_tile_z_golden.py)."""
    rb = json.load(open(os.path.join(util.GOLDEN, "run_batch_z.json")))[str(z)]
    total, gch = rb["total_slc"], rb["gene_channels"]
    assert total == tiles.state_slices(z)
    chn = total * 2
    tile = torch.arange(320 * 320 * chn, dtype=torch.float32).reshape(1, 320, 320, chn)
    gen = torch.Generator().manual_seed(rb["seed"])
    ssz = (1, 20, 20, gch)
    crd = torch.stack([torch.randint(0, ssz[k], (4000,), generator=gen) for k in range(4)])
    crd = torch.unique(crd, dim=1)
    dat = (crd[1] * 20 * gch + crd[2] * gch + crd[3] + 1).float()
    rna = torch.sparse_coo_tensor(crd, dat, ssz).to_dense()
    x, r, shape = tiles.run_batch_inputs(tile, rna, 64, 4, total, z)
    assert list(shape) == rb["shape"] and list(x.shape) == rb["imgs_shape"] and list(r.shape) == rb["rna_shape"]
    assert x[:, 1, 5, 7].long().tolist() == rb["imgs_probe"] and float(x.double().sum()) == rb["imgs_sum"]
    assert int((r != 0).sum()) == rb["rna_nonzero"] and float(r.double().sum()) == rb["rna_sum"]
    for n, probe in zip(rb["rna_probe_rows"], rb["rna_probe"]):
        assert [int(v) for v in r[n].nonzero()[:3].reshape(-1).tolist()] + [float(r[n].max())] == probe
    assert torch.equal(tc.zchunk_state(tile, total, z), tiles.zchunk_state(tile, total, z))
    assert torch.equal(tc.zchunk_rna(rna, z), tiles.zchunk_rna(rna, z))
    n, cz = shape[0], shape[1]
    out = (torch.arange(n * cz * 256 * 256, dtype=torch.float32).reshape(n, cz, 256, 256) % 2039) / 64.0
    saved = tiles.regroup_output(out, 1, 2).half()
    assert list(saved.shape[1:]) == rb["saved_shape"] and rb["saved_dtype"] == "float16"
    f = saved[0].float().reshape(-1)
    d = rb["saved_digest"]
    assert torch.equal(f[torch.tensor(d["idx"])].double(), torch.tensor(d["val"], dtype=torch.float64))
    assert abs(float(f.double().mean()) - d["mean"]) <= 1e-9


def test_sample_rejects_unclamped_sampling():
    """clip_denoised=False used to be accepted and silently ignored (the step kernel always clamps x0)."""
    from teramind_amd.diffusion import SpacedDiffusionBeatGans
    smp = SpacedDiffusionBeatGans(15, "ddim")
    with pytest.raises(NotImplementedError):
        smp.sample(model=None, shape=(1, 4, 64, 64), noise=torch.zeros(1, 4, 64, 64), clip_denoised=False)


def test_consistent_gene_provider_tiles_agree_where_they_overlap():
    """brain.consistent_gene_provider (the synthetic stand-in for gene tiles cut with overlap from one gene map,
    utils/MBADataset_tst.py:65-91; precondition of TileSweep(share_halo=True)): a tile's 2-cell halo equals its neighbours'
    interior rim, in both directions and across the corner; the z-padding slices are zero."""
    import torch
    from teramind_amd.brain import Z_PAD, consistent_gene_provider
    from teramind_amd.config import PathConfig
    cfg = PathConfig()
    g = consistent_gene_provider(cfg, "cpu", total_slc=4, density=0.05)
    a, r, d, dr = g(7, 9), g(7, 10), g(8, 9), g(8, 10)
    assert a.shape == (20, 20, (4 + 2 * Z_PAD[cfg.rna_slc]) * 500) and float(a.sum()) > 0
    assert torch.equal(a[:, 16:20], r[:, 0:4]) and torch.equal(a[16:20, :], d[0:4, :]) and torch.equal(a[16:20, 16:20], dr[0:4, 0:4])
    zp = Z_PAD[cfg.rna_slc] * 500
    assert float(a[..., :zp].abs().sum()) == 0 and float(a[..., a.shape[-1] - zp:].abs().sum()) == 0
    assert torch.equal(g(7, 9), a)                      # a pure function of the tile position
