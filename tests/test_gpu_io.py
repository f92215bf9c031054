"""-m gpu: tile I/O row -- the gene-tile scatter kernel through the C-ABI against the CPU
oracle (bit-exact: integer counts), the file-backed gene provider, and a sweep driven from gene files."""
import numpy as np
import pytest
import torch

import util
from oracle import teramind_cpu as tc
from teramind_amd import _lib, stitch
from teramind_amd.brain import GeneTileDir, TileSweep
from teramind_amd.config import PathConfig
from teramind_amd.diffusion import SpacedDiffusionBeatGans
from teramind_amd.unet import BeatGANsUNetModel

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _dense(data, crd, shape, sh, sw, gsz=20, gblk=16, zpad=500):
    c = torch.from_numpy(np.ascontiguousarray(crd.astype(np.int32))).to(DEV)
    d = torch.from_numpy(np.asarray(data).astype(np.float32)).to(DEV)
    out = torch.full((gsz, gsz, shape[2] + 2 * zpad), 7.0, device=DEV)          # must be overwritten, not accumulated into
    _lib.check(_lib.lib().tm_gene_tile_dense(_lib.ptr(c), _lib.ptr(d), d.numel(), gblk, sh, sw, gsz, shape[2], zpad,
                                             _lib.ptr(out), _lib.current_stream_ptr()), "tm_gene_tile_dense")
    return out.cpu().numpy()


@pytest.mark.parametrize("nnz,total_slc", [(0, 4), (1, 4), (777, 4), (200000, 50), (3000000, 50)])
def test_gene_tile_dense_bit_exact(nnz, total_slc):
    data, crd, shape = util.synthetic_gene_coo(7, 11, total_slc, nnz, seed=nnz)
    roi, roio = (1792, 2048, 2816, 3072), (1664, 2176, 2688, 3200)
    ref = tc.gene_tile_dense(data, crd, shape, roi, roio)
    got = _dense(data, crd, shape, -6, -6)
    assert got.shape == ref.shape and np.array_equal(got, ref)
    if nnz > 1000:
        assert ref.max() > 5         # duplicates really accumulated


def test_gene_tile_dense_other_geometry_and_bad_args():
    # rna_slc=8-style padding (1 slice) with a 32-px patch grid (gblk 8, gsz 40) and an asymmetric ROI offset
    data, crd, shape = util.synthetic_gene_coo(1, 2, 6, 50000, seed=5)
    roi, roio = (256, 512, 512, 768), (128, 640, 448, 832)
    ref = tc.gene_tile_dense(data, crd, shape, roi, roio, gblk=8, pad=32, size=256, spad=1)
    got = _dense(data, crd, shape, 4 - 128 // 8, 4 - 64 // 8, gsz=40, gblk=8)
    assert np.array_equal(got, ref)
    L = _lib.lib()
    assert L.tm_gene_tile_dense(None, None, 5, 16, -6, -6, 20, 2000, 500, None, None) != 0
    assert b"null" in L.tm_last_error()


def test_gene_tile_dir_provider_and_sweep(tmp_path):
    cfg = PathConfig()
    slc, T = 4, 2
    util.write_gene_dir(str(tmp_path / "gene"), rows=[1], cols=[2, 3], total_slc=slc, nnz=30000)
    prov = GeneTileDir(tmp_path / "gene", cfg, DEV, total_slc=slc)
    dense = {}
    for col in (2, 3):
        data, crd, shape = util.synthetic_gene_coo(1, col, slc, 30000)
        roi = (256, 512, col * 256, col * 256 + 256)
        roio = (128, 640, col * 256 - 128, col * 256 + 384)
        dense[(1, col)] = torch.from_numpy(tc.gene_tile_dense(data, crd, shape, roi, roio))
        assert torch.equal(prov(1, col).cpu(), dense[(1, col)])
        assert torch.equal(prov(1, col).cpu(), dense[(1, col)])          # second call: resident COO, same result
    with pytest.raises(FileNotFoundError):
        prov(5, 5)
    sd = util.state_dict(cfg)
    model = BeatGANsUNetModel(cfg, DEV).load_state_dict(sd)
    kw = dict(hst=256, wst=512, hnm=1, wnm=2, total_epochs=T, total_slc=slc, device=DEV, batch_tiles=2)
    a = TileSweep(cfg, SpacedDiffusionBeatGans(T, "ddim"), model, prov, **kw).test()
    b = TileSweep(cfg, SpacedDiffusionBeatGans(T, "ddim"), model, lambda r, c: dense[(r, c)].to(DEV), **kw).test()
    assert torch.equal(a, b)
    # export -> stitch from files == stitch of the resident state
    sw = TileSweep(cfg, SpacedDiffusionBeatGans(T, "ddim"), model, prov, **kw)
    sw.test()
    d = sw.save_step(tmp_path / "out", compressor="zlib")
    m = stitch.stitch_dir(d, 256, 512, 1, 2, slc)
    assert np.array_equal(m, stitch.stitch_state(sw.local_state(), slc).cpu().numpy())


def test_stitch_state_on_cuda_canvas_equals_oracle_and_step_files(tmp_path):
    """Row f2 on the device: stitch.stitch_state on the RESIDENT (CUDA) canvas of a sweep == the oracle's per-tile stitch
    (`gen_col` of infer_brn.py:57-86: '(c s) h w -> (s c) h w', ((g + 1) * 127.5) in float16 -> uint8) of the same state
    == stitch_dir of the tile files save_step writes, for both canvas layouts."""
    cfg = PathConfig(compute_dtype="bf16")
    slc, T = 4, 2
    from teramind_amd.brain import consistent_gene_provider
    model = BeatGANsUNetModel(cfg, DEV).load_state_dict(util.state_dict(cfg))
    genes = consistent_gene_provider(cfg, DEV, total_slc=slc, density=0.05)
    for state in ("fp16", "fp32x2"):
        sw = TileSweep(cfg, SpacedDiffusionBeatGans(T, "ddim"), model, genes, hst=512, wst=768, hnm=2, wnm=2, total_epochs=T,
                       total_slc=slc, device=DEV, batch_tiles=2, init="device", state=state)
        sw.test()
        st = sw.local_state()
        assert st.is_cuda
        got = stitch.stitch_state(st, slc)
        assert got.is_cuda and got.dtype == torch.uint8 and got.shape == (2 * slc, 512, 512)
        h = st.half().cpu().numpy()
        for ph in range(2):
            for pw in range(2):
                ref = tc.stitch_tile_uint8(h[:, ph * 256:(ph + 1) * 256, pw * 256:(pw + 1) * 256], slc)
                assert np.array_equal(got[:, ph * 256:(ph + 1) * 256, pw * 256:(pw + 1) * 256].cpu().numpy(), ref)
        d = sw.save_step(tmp_path / ("out_" + state), compressor="zlib")
        assert np.array_equal(stitch.stitch_dir(d, 512, 768, 2, 2, slc), got.cpu().numpy())


def test_share_halo_guard_on_gene_tile_dirs(tmp_path):
    """TileSweep(share_halo=True) on real gene tile directories: tiles cut from ONE gene map (synth.write_gene_tile_dir) agree
    on their overlap and the window sweep equals the per-tile sweep bit for bit; a directory in which one tile disagrees with
    its neighbour in a single shared cell is refused with the tiles named (utils/MBADataset_tst.py:65-89: every tile brings
    its own halo, so dropping one tile's version would silently change the result)."""
    from teramind_amd import synth
    cfg = PathConfig(compute_dtype="bf16")
    slc, T = 4, 2
    model = BeatGANsUNetModel(cfg, DEV).load_state_dict(util.state_dict(cfg))
    synth.write_gene_tile_dir(str(tmp_path / "good"), 2, 2, 1500, hst=256, wst=512, total_slc=slc)
    synth.write_gene_tile_dir(str(tmp_path / "bad"), 2, 2, 1500, hst=256, wst=512, total_slc=slc, spoil=(1, 2))
    good = GeneTileDir(tmp_path / "good", cfg, DEV, total_slc=slc)
    a, b = good(1, 2), good(1, 3)
    assert float(a.sum()) > 0 and torch.equal(a[:, 16:20], b[:, 0:4]) and torch.equal(good(1, 2)[16:20], good(2, 2)[0:4])
    kw = dict(hst=256, wst=512, hnm=2, wnm=2, total_epochs=T, total_slc=slc, device=DEV, init="device", state="fp16")
    per_tile = TileSweep(cfg, SpacedDiffusionBeatGans(T, "ddim"), model, good, batch_tiles=1, **kw).test()
    window = TileSweep(cfg, SpacedDiffusionBeatGans(T, "ddim"), model, good, batch_tiles=2, batch_rows=2, share_halo=True, **kw).test()
    assert torch.equal(per_tile, window)
    bad = GeneTileDir(tmp_path / "bad", cfg, DEV, total_slc=slc)
    with pytest.raises(ValueError, match=r"share_halo: gene tile \(row 1, col 2\) and tile \(row 1, col 3\) disagree"):
        TileSweep(cfg, SpacedDiffusionBeatGans(T, "ddim"), model, bad, batch_tiles=2, batch_rows=2, share_halo=True, **kw).test()
    # per-tile calls remain available for such data
    TileSweep(cfg, SpacedDiffusionBeatGans(T, "ddim"), model, bad, batch_tiles=2, **kw).test()


@pytest.mark.parametrize("name", ["interior", "roi_corner", "asym_blk8", "spad3_blk16"])
def test_gene_tile_dense_vs_reference_pad_gn(name):
    """tm_gene_tile_dense (block sum + halo shift + crop + z padding in one scatter pass) against what the reference's OWN
    MBADataset_tst._pad_gn returned for the same entries (tests/golden/io_ref_pad.npz, oracle/make_io_ref_golden.py)."""
    import os
    from teramind_amd import formats
    z = np.load(os.path.join(util.GOLDEN, "io_ref_pad.npz"))
    gblk, pad, size, spad, slc, H, W = (int(v) for v in z[f"gn/{name}/params"])
    r = [int(v) for v in z[f"gn/{name}/roi"]]
    sh, sw = formats.gene_tile_shift(r[:4], r[4:], gblk, pad)
    ref = np.zeros(tuple(int(v) for v in z[f"gn/{name}/out_ssz"]), dtype=np.float32)
    crd = z[f"gn/{name}/out_crd"]
    np.add.at(ref, (crd[0], crd[1], crd[2]), z[f"gn/{name}/out_dat"].astype(np.float32))
    got = _dense(z[f"gn/{name}/data"], z[f"gn/{name}/pix"], (H, W, slc * 500), sh, sw, gsz=(size + 2 * pad) // gblk, gblk=gblk, zpad=spad * 500)
    assert got.shape == ref.shape and np.array_equal(got, ref) and ref.sum() > 0
