"""CPU tests of the tile I/O row: container formats, the Blosc decoder of the C-ABI
library against frames produced by the real C-Blosc (tests/golden/io_*; oracle/make_io_golden.py), the gene-tile
oracle against hand-computed answers, stitching, and step-directory save / resume of the sweep."""
import json
import os
import zipfile

import numpy as np
import pytest
import torch

import util
from oracle import teramind_cpu as tc
from teramind_amd import formats, stitch, tiles
from teramind_amd.brain import TileSweep
from teramind_amd.config import PathConfig


# ---- Blosc / zarr ---------------------------------------------------------------------------------
def test_blosc_decoder_against_c_blosc_frames():
    z = np.load(os.path.join(util.GOLDEN, "io_blosc_frames.npz"))
    names = sorted({k.split("/")[0] for k in z.files if "/" in k})
    assert len(names) >= 10
    for n in names:
        assert formats._blosc_decode(z[n + "/frame"].tobytes()) == z[n + "/plain"].tobytes(), n


def test_blosc_decoder_rejects_damaged_frames():
    z = np.load(os.path.join(util.GOLDEN, "io_blosc_frames.npz"))
    fr = z["f16_small_default/frame"].tobytes()
    with pytest.raises(RuntimeError, match="blosc"):
        formats._blosc_decode(fr[:10])                        # shorter than the header
    with pytest.raises(RuntimeError, match="truncated"):
        formats._blosc_decode(fr[: len(fr) // 2])
    bad = bytearray(fr)
    bad[2] = (bad[2] & 0x1F) | (4 << 5)                       # codec id zstd
    with pytest.raises(RuntimeError, match="lz4"):
        formats._blosc_decode(bytes(bad))
    bad = bytearray(fr)
    bad[40:60] = b"\xff" * 20                                 # garbage inside the first lz4 stream
    with pytest.raises(RuntimeError):
        formats._blosc_decode(bytes(bad))


def test_zarr_zip_reader_on_blosc_fixture():
    a = formats.read_state_tile(os.path.join(util.GOLDEN, "io_state_tile_blosc.zip"))
    e = np.load(os.path.join(util.GOLDEN, "io_state_tile_expected.npy"))
    assert a.dtype == np.float16 and np.array_equal(a, e)


@pytest.mark.parametrize("comp", [None, "zlib"])
def test_state_tile_round_trip(tmp_path, comp):
    t = torch.randn(100, 256, 256).clamp(-1, 1).half().numpy()
    p = tmp_path / "5120_5376_7680_7936.zip"
    formats.write_state_tile(p, t, comp)
    assert np.array_equal(formats.read_state_tile(p), t)
    assert formats.parse_state_tile_name(p) == (5120, 5376, 7680, 7936)
    with zipfile.ZipFile(p) as z:                              # the layout zarr's ZipStore expects
        assert set(z.namelist()) == {".zarray", "0.0.0"}
        meta = json.loads(z.read(".zarray"))
        assert meta["zarr_format"] == 2 and meta["dtype"] == "<f2" and meta["shape"] == [100, 256, 256]
        assert all(i.compress_type == zipfile.ZIP_STORED for i in z.infolist())


def test_zarr_reader_edge_chunks_missing_chunks_and_fill(tmp_path):
    a = np.arange(5 * 7, dtype="<i4").reshape(5, 7)
    meta = {"chunks": [2, 4], "compressor": None, "dtype": "<i4", "fill_value": -3, "filters": None, "order": "C",
            "shape": [5, 7], "zarr_format": 2}
    p = tmp_path / "a.zip"
    with zipfile.ZipFile(p, "w") as z:
        z.writestr(".zarray", json.dumps(meta))
        for i in range(3):
            for j in range(2):
                if (i, j) == (1, 1):
                    continue                                   # absent chunk -> fill value
                blk = np.full((2, 4), 99, dtype="<i4")         # edge chunks are stored at full chunk size
                sub = a[2 * i:2 * i + 2, 4 * j:4 * j + 4]
                blk[:sub.shape[0], :sub.shape[1]] = sub
                z.writestr(f"{i}.{j}", blk.tobytes())
    e = a.copy()
    e[2:4, 4:7] = -3
    assert np.array_equal(formats.read_zarr_zip(p), e)
    with zipfile.ZipFile(tmp_path / "b.zip", "w") as z:
        z.writestr("x", b"")
    with pytest.raises(ValueError, match="zarray"):
        formats.read_zarr_zip(tmp_path / "b.zip")


# ---- gene tiles -------------------------------------------------------------------------------------
def test_gene_npz_round_trip_and_names(tmp_path):
    data, crd, shape = util.synthetic_gene_coo(3, 4, total_slc=4, nnz=500)
    p = tmp_path / "768_1024_1024_1280_640_1152_896_1408.npz"
    formats.write_gene_npz(p, data, crd, shape)
    d2, c2, s2 = formats.read_gene_npz(p)
    assert np.array_equal(d2, data) and np.array_equal(c2, crd) and s2 == shape
    with np.load(p) as z:                                       # the four arrays sparse.load_npz reads
        assert set(z.files) == {"data", "coords", "shape", "fill_value"}
    roi, roio = formats.parse_gene_tile_name(p)
    assert roi == (768, 1024, 1024, 1280) and roio == (640, 1152, 896, 1408)
    assert formats.gene_tile_shift(roi, roio) == (-6, -6)
    assert os.path.basename(str(p))[:-4] == tiles.gene_tile_names(hst=768, wst=1024, hnm=1, wnm=1)[0]
    with pytest.raises(ValueError):
        formats.parse_gene_tile_name("1_2_3.npz")


def test_oracle_gene_tile_known_answers():
    """Hand-computed: gblk 16, pad 32, roi-roio = 128 -> cell = px//16 - 6, kept if 0 <= cell < 20;
    channel c -> c + 500 (one padding slice either side)."""
    roi, roio = (256, 512, 256, 512), (128, 640, 128, 640)
    crd = np.array([[96, 111, 96, 95, 415, 416, 200, 200, 207],        # h
                    [96, 100, 112, 300, 415, 100, 331, 331, 335],       # w
                    [0, 0, 7, 3, 24999, 5, 600, 600, 600]])             # channel
    dat = np.array([1, 2, 4, 8, 16, 32, 3, 5, 7], dtype=np.uint8)
    out = tc.gene_tile_dense(dat, crd, (512, 512, 25000), roi, roio)
    assert out.shape == (20, 20, 26000) and out.dtype == np.float32
    exp = {(0, 0, 500): 3.0,            # (96,96) and (111,100) share cell 6-6=0 / 0
           (0, 1, 507): 4.0,            # w=112 -> cell 7-6
           (19, 19, 25499): 16.0,       # 415//16 = 25 -> 19 (last kept cell)
           (6, 14, 1100): 15.0}         # 200//16=12->6 and 207//16=12; 331//16=20->14, 335//16=20: 3+5+7
    # h=95 -> cell -1 and h=416 -> cell 20 are cropped
    got = {tuple(int(i) for i in idx): float(out[tuple(idx)]) for idx in np.argwhere(out)}
    assert got == exp
    assert out[:, :, :500].sum() == 0 and out[:, :, 25500:].sum() == 0


# ---- stitcher -----------------------------------------------------------------------------------------
def test_stitch_matches_oracle_and_dir_round_trip(tmp_path):
    g = torch.Generator().manual_seed(3)
    slc, hnm, wnm = 3, 2, 2
    state = torch.rand((2 * slc, hnm * 256, wnm * 256), generator=g) * 2 - 1
    state[0, 0, :4] = torch.tensor([-1.0, 1.0, 0.0, 0.999])
    state = state.half()
    got = stitch.stitch_state(state, slc)
    assert got.dtype == torch.uint8 and got.shape == state.shape
    for ph in range(hnm):
        for pw in range(wnm):
            t = state[:, ph * 256:(ph + 1) * 256, pw * 256:(pw + 1) * 256]
            assert np.array_equal(got[:, ph * 256:(ph + 1) * 256, pw * 256:(pw + 1) * 256].numpy(),
                                  tc.stitch_tile_uint8(t.numpy(), slc))
            formats.write_state_tile(tmp_path / (tiles.state_tile_name(4 + ph, 9 + pw) + ".zip"), t.numpy())
    assert got[0, 0, 0] == 0 and got[0, 0, 1] == 255 and got[0, 0, 2] == 127
    # channel (c, s) lands at index s * 2 + c
    assert torch.equal(got[1 * 2 + 1], stitch.to_uint8(state[1 * slc + 1]))
    m = stitch.stitch_dir(tmp_path, 4 * 256, 9 * 256, hnm, wnm, slc)
    assert np.array_equal(m, got.numpy())
    stitch.save_slices(m[:2], tmp_path / "img", names=[0, 1])
    from PIL import Image
    assert np.array_equal(np.asarray(Image.open(tmp_path / "img" / "all_1.tif")), m[1])


# ---- sweep: save / resume through reference-format step directories --------------------------------------
class _ToySampler:
    """Deterministic stand-in with the sampler.sample signature: x_{t-1} = 0.5 * centre crop + mean gene term."""

    def sample(self, model=None, shape=None, imgs=None, noise=None, r_start=None, patch_size=64, idx=None, **kw):
        n, c, H, W = shape
        P1, P2 = H // patch_size + 1, W // patch_size + 1
        x = imgs.reshape(n, P1, P2, c, patch_size, patch_size).permute(0, 3, 1, 4, 2, 5).reshape(n, c, P1 * patch_size, P2 * patch_size)
        h = patch_size // 2
        g = r_start.reshape(n, -1).mean(1).reshape(n, 1, 1, 1)
        return 0.5 * x[:, :, h:-h, h:-h] + 0.01 * (idx + 1) + g


@pytest.mark.parametrize("state", ["fp32x2", "fp16"])
def test_sweep_save_and_resume_from_step_dir(tmp_path, state):
    cfg = PathConfig()
    slc, T = 4, 3
    genes = lambda r, c: torch.full((20, 20, (slc + 2) * 500), float(r + c) * 1e-3)
    kw = dict(hst=512, wst=256, hnm=2, wnm=2, total_epochs=T, total_slc=slc, state=state)
    a = TileSweep(cfg, _ToySampler(), None, genes, **kw)
    a.step()
    a.step()
    d = a.save_step(tmp_path / "out")
    assert d.endswith("out_2") and len(os.listdir(d)) == 4
    assert os.path.exists(os.path.join(d, "512_768_256_512.zip"))
    a.step()
    b = TileSweep(cfg, _ToySampler(), None, genes, **kw)
    b.load_step(tmp_path / "out", 2)
    assert b.epoch == 2
    b.step()
    assert torch.equal(a.local_state(), b.local_state())       # state is fp16-representable after every step
    with pytest.raises(ValueError):
        TileSweep(cfg, _ToySampler(), None, genes, **kw).save_step(tmp_path / "x")


# ---- the reference's own _pad_gn / _pad_im(step > 0) (oracle/make_io_ref_golden.py) ------------------------------------
REF_PAD = np.load(os.path.join(util.GOLDEN, "io_ref_pad.npz"))
GN_CASES = ["interior", "roi_corner", "asym_blk8", "spad3_blk16"]


def ref_pad_gn_dense(name):
    """Dense [gsz, gsz, C] tensor of the COO triple the reference's `_pad_gn` returned (duplicates add, as to_dense does)."""
    dat, crd, ssz = REF_PAD[f"gn/{name}/out_dat"], REF_PAD[f"gn/{name}/out_crd"], REF_PAD[f"gn/{name}/out_ssz"]
    out = np.zeros(tuple(int(v) for v in ssz), dtype=np.float32)
    np.add.at(out, (crd[0], crd[1], crd[2]), dat.astype(np.float32))
    return out


@pytest.mark.parametrize("name", GN_CASES)
def test_gene_tile_shift_and_crop_vs_reference_pad_gn(name):
    """oracle.gene_tile_dense and formats.gene_tile_shift against MBADataset_tst._pad_gn ITSELF (utils/MBADataset_tst.py:80-89,
    run on a stand-in COO object): interior tile, padded ROI clipped at the slide corner, asymmetric offset with 8-px
    cells, 3-slice z padding."""
    gblk, pad, size, spad, slc, H, W = (int(v) for v in REF_PAD[f"gn/{name}/params"])
    r = [int(v) for v in REF_PAD[f"gn/{name}/roi"]]
    roi, roio = r[:4], r[4:]
    pix, data = REF_PAD[f"gn/{name}/pix"], REF_PAD[f"gn/{name}/data"]
    ref = ref_pad_gn_dense(name)
    got = tc.gene_tile_dense(data, pix, (H, W, slc * 500), roi, roio, gblk=gblk, pad=pad, size=size, spad=spad)
    assert got.shape == ref.shape and np.array_equal(got, ref) and ref.sum() > 0
    # the product's shift rule: every kept entry of the reference sits at cell (pixel // gblk + shift)
    sh, sw = formats.gene_tile_shift(roi, roio, gblk, pad)
    gsz = (size + 2 * pad) // gblk
    ch, cw = pix[0] // gblk + sh, pix[1] // gblk + sw
    keep = (ch >= 0) & (ch < gsz) & (cw >= 0) & (cw < gsz)
    mine = np.zeros_like(ref)
    np.add.at(mine, (ch[keep], cw[keep], pix[2][keep] + spad * 500), data[keep].astype(np.float32))
    assert np.array_equal(mine, ref)


def _ref_state_tile(row, col, chn, size=256):
    """oracle/make_io_ref_golden.state_tile: what the stubbed zarr.load returned for tile (row, col)."""
    c, h, w = np.meshgrid(np.arange(chn), np.arange(size) // 32, np.arange(size) // 32, indexing="ij")
    return (((row * 5 + col) * 64 + c * 8 + h) / 64.0 - 2.0 + w / 1024.0).astype(np.float16)


@pytest.mark.parametrize("state", ["fp32x2", "fp16"])
def test_halo_assembly_at_later_steps_vs_reference_pad_im(state):
    """TileSweep._window at epoch > 0 (the 320 x 320 padded tile cut from the resident canvas) == the reference's
    MBADataset_tst._pad_im(roi, 2) (utils/MBADataset_tst.py:91-123) reading the previous step's 3 x 3 neighbour tiles --
    centre, corner and edge tiles of a 3 x 3 ROI (outside the ROI: -1), both canvas layouts."""
    chn, hst, wst, hnm, wnm, step = (int(v) for v in REF_PAD["im/params"])
    # (3 slices x 2 stains: the per-slice model, rna_slc 1, takes any slice count; the window assembly does not depend on it)
    sw = TileSweep(PathConfig(rna_slc=1), sampler=None, model=None, gene_provider=None, hst=hst * 256, wst=wst * 256, hnm=hnm,
                   wnm=wnm, total_epochs=15, total_slc=chn // 2, state=state)
    assert sw.chn == chn
    for lr in range(hnm):
        for c in range(wnm):
            sw._centre(sw.cur, lr, c).copy_(torch.from_numpy(_ref_state_tile(hst + lr, wst + c, chn).astype(np.float32)))
    sw.epoch = step
    sw._exchange(sw.cur)
    for key in [k for k in REF_PAD.files if k.startswith("im/") and k != "im/params"]:
        lr, c = (int(v) for v in key[3:].split("_"))
        ref = torch.from_numpy(REF_PAD[key].astype(np.float32))
        assert torch.equal(sw._window(lr, c).float(), ref), key
