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
