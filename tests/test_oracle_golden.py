"""CPU: the oracle (oracle/teramind_cpu.py) against the golden vectors minted from the REAL
reference by oracle/make_golden.py -- this is what pins the oracle on machines where
/root/reference does not exist (the GPU box)."""
import json
import os

import numpy as np
import pytest
import torch

import util
from oracle import teramind_cpu as tc
from teramind_amd import synth
from teramind_amd.config import PathConfig

G = util.GOLDEN
INT = json.load(open(os.path.join(G, "integer_paths.json")))


def test_space_timesteps_bit_exact():
    for key, rec in INT["space_timesteps"].items():
        assert sorted(tc.space_timesteps(rec["T"], rec["section_counts"])) == rec["steps"], key


def test_timestep_maps_and_tables_bit_exact():
    tabs = np.load(os.path.join(G, "tables.npz"))
    for T, gen in [(15, "ddim"), (50, "ddim"), (50, "ddpm"), (1000, "ddpm")]:
        sch = tc.make_schedule(T, gen)
        if f"{gen}{T}" in INT["timestep_map"]:
            assert sch.timestep_map == INT["timestep_map"][f"{gen}{T}"]
        for name in ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
                     "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
                     "posterior_mean_coef1", "posterior_mean_coef2"]:
            assert np.array_equal(getattr(sch, name), tabs[f"{gen}{T}/{name}"]), (gen, T, name)


def test_sparse_repatch_bit_exact():
    r = INT["sparse_repatch"]
    crd, ssz = tc.sparse_repatch(torch.tensor(r["crd_in"]), r["ssz"], r["sz"])
    assert crd.tolist() == r["crd_out"] and list(ssz) == r["ssz_out"]


def test_lcg_and_names():
    for k, v in INT["lcg"].items():
        assert tc.lcg(int(k)) == v
    assert [n + ".npz" for n in tc.gene_tile_names(hst=256, wst=512, hnm=2, wnm=3)] == INT["gn_sublst"]


@pytest.mark.parametrize("b,P,seed", [(1, 1, 0), (1, 2, 3)])
def test_unet_forward_vs_reference_fixture(b, P, seed):
    gold = np.load(os.path.join(G, "unet_full.npz"))
    cfg = PathConfig()
    sd = util.state_dict(cfg)
    p = P + 1
    ne = b * p * p
    x = synth.normal("x", (ne, 4, 64, 64), seed)
    rna = synth.gene_counts("rna", (ne, 4, 4, 2000), seed)
    t = torch.tensor([(137 * (i + 1) + 61 * seed) % 1000 for i in range(b)], dtype=torch.long)
    with torch.inference_mode():
        pred, pred2 = tc.unet_forward(sd, tc.oracle_config_from(cfg), x, t, rna, p, p, want_pred2=(P == 1))
    tag = f"b{b}_P{P}_s{seed}"
    # fp32 re-association noise between two CPU evaluation orders (measured 1.5e-6)
    assert torch.allclose(pred, torch.from_numpy(gold[f"{tag}/pred"]), atol=2e-5, rtol=0)
    if P == 1:
        assert torch.allclose(pred2, torch.from_numpy(gold[f"{tag}/pred2"]), atol=2e-5, rtol=0)


@pytest.mark.parametrize("dtype,lo,hi", [("bf16", 4e-3, 1e-2), ("f16", 5e-4, 1.3e-3)])
def test_autocast_reference_fixture_is_consistent(dtype, lo, hi):
    """tests/golden/unet_autocast_ref.npz (the reference under torch.autocast('cpu', <dtype>), oracle/make_autocast_golden.py)
    belongs to the same inputs as unet_full.npz: its distance to the fp32 fixture is the stored one and is the size a 16-bit
    evaluation of this network has (what the GPU test compares the HIP 16-bit modes with)."""
    gold, full = np.load(os.path.join(G, "unet_autocast_ref.npz")), np.load(os.path.join(G, "unet_full.npz"))
    for tag in ("b1_P1_s0", "b1_P2_s3"):
        a, r = torch.from_numpy(gold[f"{dtype}/{tag}/pred"]), torch.from_numpy(full[f"{tag}/pred"])
        assert a.shape == r.shape and torch.isfinite(a).all()
        rel = ((a - r).pow(2).mean().sqrt() / r.pow(2).mean().sqrt()).item()
        assert abs(rel - float(gold[f"{dtype}/{tag}/rel_l2_vs_fp32"])) < 1e-9
        assert lo < rel < hi, (dtype, tag, rel)


def test_sampler_trajectories_vs_reference_fixture():
    gold = np.load(os.path.join(G, "sampler_traj.npz"))
    cfg = PathConfig()
    oc = tc.oracle_config_from(cfg)
    sd = util.state_dict(cfg)
    rna = synth.gene_counts("traj/rna", (4, 4, 4, 2000), 0)
    # mode A, DDPM T=3.  RNG draw order of the reference: x_T, placeholder imgs, then one per step
    assert int(gold["modeA_ddpm3/ndraws"][0]) == 5
    xT = synth.normal("traj/ddpm3/0", (1, 4, 64, 64), 0)
    noises = [synth.normal(f"traj/ddpm3/{k + 2}", (4, 4, 64, 64), 0) for k in range(3)]
    with torch.inference_mode():
        out = tc.sample_loop(sd, oc, tc.make_schedule(3, "ddpm"), "ddpm", xT, rna, noises)
    # first step sits at t=999 of a 3-step schedule: sqrt(1/abar - 1) ~ 158 amplifies the 1.5e-6 forward noise
    assert torch.allclose(out, torch.from_numpy(gold["modeA_ddpm3/final"]), atol=2e-3, rtol=0)
    # mode B single step, DDIM-15, idx 7, incl. the fp16 cast of test_brn.py:222
    xp = synth.normal("traj/modeB/x", (4, 4, 64, 64), 0) * 0.8
    sch = tc.make_schedule(15, "ddim")
    with torch.inference_mode():
        pred, _ = tc.unet_forward(sd, oc, xp, torch.tensor([sch.timestep_map[7]]), rna, 2, 2)
        out = tc.sampler_step(sch, "ddim", xp, pred, 7, 1, 1)
    assert torch.allclose(out, torch.from_numpy(gold["modeB_ddim15_idx7/out"]), atol=2e-5, rtol=0)
    half = torch.from_numpy(gold["modeB_ddim15_idx7/out_half"].astype(np.float32))
    assert (out.half().float() - half).abs().max() <= 2e-3          # at most one fp16 ulp at |x| <= 2


def test_attention_maps_vs_reference_fixture():
    gold = np.load(os.path.join(G, "attn_maps.npz"))
    cfg = PathConfig()
    sdv = util.state_dict(cfg, vis_only=True)
    rna = synth.gene_counts("rna_vis", (2, 4, 4, 2000), 1, density=0.05)
    with torch.inference_mode():
        attn, mid = tc.gene_attention_maps(sdv, tc.oracle_config_from(cfg), rna)
    assert torch.allclose(attn[:, 0], torch.from_numpy(gold["attn_b0"]), atol=1e-7, rtol=1e-5)
    assert torch.equal(mid, torch.from_numpy(gold["mid"]))
    assert torch.allclose(attn[:, :, [75, 191]][:, :, :, [75, 191]], torch.from_numpy(gold["attn_glst"]), atol=1e-7, rtol=1e-5)


def test_attention_driver_readout_vs_reference_fixture():
    """K18 (test_attn.py:404-431): the fixture is what the reference's own Tester._run_batch saved."""
    gold = np.load(os.path.join(G, "attn_readout.npz"))["out"]
    cfg = PathConfig()
    tile = synth.gene_counts("attn/tile", (1, 20, 20, 26000), 0, density=0.05)
    with torch.inference_mode():
        out = tc.attn_tile_readout(util.state_dict(cfg, vis_only=True), tc.oracle_config_from(cfg), tile, [75, 191])
    assert out.dtype == torch.float16 and tuple(out.shape[1:]) == gold.shape
    assert torch.equal(out[0].float(), torch.from_numpy(gold.astype(np.float32)))


# ---- the other configuration-surface points -------------------------------------
@pytest.mark.parametrize("case", __import__("config_cases").CONFIGS[:5] + [(64, 1, "all", 81)], ids=lambda c: __import__("config_cases").tag_of(c))
def test_oracle_vs_reference_other_configs(case):
    """oracle/teramind_cpu.py against outputs minted from the REAL reference (oracle/make_config_golden.py) for
    patch_size 32 / 128, rna_slc 1 / 8 / 16, a single stain and the 81-gene M2H subset (a CPU subset of the list the
    GPU suite covers; each forward takes seconds)."""
    import config_cases as cc
    from oracle import teramind_cpu as tc
    from teramind_amd.weights import hashed_state_dict
    gold = np.load(os.path.join(util.GOLDEN, "unet_configs.npz"))
    cfg = cc.path_config(case)
    x, rna, t = cc.inputs(cfg)
    torch.set_num_threads(8)
    with torch.inference_mode():
        pred, pred2 = tc.unet_forward(hashed_state_dict(cfg, 0), tc.oracle_config_from(cfg), x, t, rna, 2, 2, want_pred2=True)
    for name, got in (("pred", pred), ("pred2", pred2)):
        ref = gold[f"{cc.tag_of(case)}/{name}"]
        assert np.abs(cc.digest(got) - ref).max() < 5e-5, (case, name)


# ---- training objective, forward half ---------------------------------------------
def test_oracle_training_loss_vs_reference():
    """oracle.training_losses against the loss the reference's own GaussianDiffusionBeatGans.training_losses
    returned (eval mode, seeded crop, given noise; oracle/make_train_golden.py)."""
    import train_cases as trc
    from teramind_amd.weights import hashed_state_dict
    gold = np.load(os.path.join(util.GOLDEN, "train_loss.npz"))
    cfg = PathConfig()
    sd = hashed_state_dict(cfg, 0)
    sch = tc.make_schedule(1000, "ddpm")
    torch.set_num_threads(8)
    name, (seed, loss_type) = "mse_seed3", trc.CASES["mse_seed3"]
    x_pad, rna, imgs, t, pos, mask, idx, noise = trc.make_inputs(seed)
    ix, iy = (int(v) for v in gold[f"{name}/crop"])
    with torch.inference_mode():
        loss, x_t = tc.training_losses(sd, tc.oracle_config_from(cfg), sch, x_pad, rna, t, mask, noise, ix, iy, 64, loss_type)
    assert abs(float(loss) - float(gold[f"{name}/loss"])) < 2e-5 * float(gold[f"{name}/loss"])
    st = gold[f"{name}/x_t_stats"]
    assert abs(x_t.double().mean().item() - st[0]) < 1e-7 and abs(x_t.double().abs().max().item() - st[1]) < 1e-6
