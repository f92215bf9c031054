#!/usr/bin/env python3
"""Mints tests/golden/io_ref_pad.npz by running the REFERENCE's own tile-loader functions  (TEST INFRASTRUCTURE).

`utils/MBADataset_tst.py` cannot be used as a whole here (its `__getitem__` needs the real `sparse` / `zarr`
packages), but two of its functions touch nothing of those packages' arithmetic:
  * `MBADataset_tst._pad_gn` (:80-89) reads only `.data / .coords / .shape` of its argument: a stand-in object with
    numpy arrays suffices.  It is the halo shift + crop of the gene tile -- what `oracle.gene_tile_dense` and
    `formats.gene_tile_shift` restate after their own block sum;
  * `MBADataset_tst._pad_im(roi, step > 0)` (:91-123) assembles the 320 x 320 padded tile from the 3 x 3 neighbour
    tiles of the previous step through `zarr.load(path)`: with `zarr.load` replaced by a function that returns seeded
    arrays keyed by the requested file name, its slicing (incl. ROI-border tiles, filled with -1) is the reference's.
The script imports the reference with empty `zarr` / `sparse` stand-in modules (oracle/ref_harness.py), calls the two
functions, and stores inputs + outputs.  Run once in this container:  python oracle/make_io_ref_golden.py"""
import contextlib
import importlib
import io
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def state_tile(row: int, col: int, chn: int, size: int = 256) -> np.ndarray:
    """The stand-in for a step directory's '{r0}_{r1}_{c0}_{c1}.zip' tile: float16 [chn, size, size], exactly
    representable values that identify (tile, channel, 32-px block) -- tests regenerate it with the same formula."""
    c, h, w = np.meshgrid(np.arange(chn), np.arange(size) // 32, np.arange(size) // 32, indexing="ij")
    return (((row * 5 + col) * 64 + c * 8 + h) / 64.0 - 2.0 + w / 1024.0).astype(np.float16)


def main():
    from oracle import ref_harness
    ref_harness.load()                                   # import stubs + /root/reference on sys.path
    with contextlib.redirect_stdout(io.StringIO()):
        ds = importlib.import_module("utils.MBADataset_tst")
    out = {}

    # ---- _pad_gn on stand-in COO objects: (name, gblk, pad, size, spad, roi, roio, H, W of the file's padded ROI in px) ----
    cases = [("interior", 16, 32, 256, 1, (256, 512, 512, 768), (128, 640, 384, 896)),
             ("roi_corner", 16, 32, 256, 1, (0, 256, 0, 256), (0, 384, 0, 384)),               # padded ROI clipped at the slide's edge
             ("asym_blk8", 8, 32, 256, 1, (256, 512, 512, 768), (128, 640, 448, 832)),
             ("spad3_blk16", 16, 32, 256, 3, (512, 768, 256, 512), (384, 896, 256, 640))]
    rng = np.random.default_rng(11)
    for name, gblk, pad, size, spad, roi, roio in cases:
        H, W, slc = roio[1] - roio[0], roio[3] - roio[2], 3
        nnz = 400
        pix = np.stack([rng.integers(0, H, nnz), rng.integers(0, W, nnz), rng.integers(0, slc * 500, nnz)]).astype(np.int64)
        pix[:, :40] = pix[:, 40:80]                                        # repeated coordinates
        data = rng.integers(1, 6, nnz).astype(np.uint16)
        # what _getgene hands to _pad_gn: cell coordinates (gblk x gblk block sum), channels shifted by spad * 500
        cells = np.stack([pix[0] // gblk, pix[1] // gblk, pix[2] + spad * 500])
        gn = types.SimpleNamespace(data=data.copy(), coords=cells.copy(), shape=(H // gblk, W // gblk, slc * 500 + spad * 1000))
        d = ds.MBADataset_tst.__new__(ds.MBADataset_tst)
        d.gblk, d.gsz, d.psz = gblk, (size + 2 * pad) // gblk, pad // gblk
        dat, crd, ssz = d._pad_gn(gn, np.array(roi), np.array(roio))
        out[f"gn/{name}/params"] = np.array([gblk, pad, size, spad, slc, H, W], dtype=np.int64)
        out[f"gn/{name}/roi"] = np.array(roi + roio, dtype=np.int64)
        out[f"gn/{name}/pix"] = pix
        out[f"gn/{name}/data"] = data
        out[f"gn/{name}/out_dat"] = np.asarray(dat)
        out[f"gn/{name}/out_crd"] = np.asarray(crd)
        out[f"gn/{name}/out_ssz"] = np.array(ssz, dtype=np.int64)
        print(f"_pad_gn {name}: {nnz} -> {len(dat)} entries, ssz {ssz}")

    # ---- _pad_im(roi, 2) with zarr.load stubbed: 3 x 3 ROI at tile rows 2.., cols 3.. ----
    chn, hst, wst, hnm, wnm, step = 6, 2, 3, 3, 3, 2
    asked = []

    def fake_load(pth):
        nm = os.path.basename(str(pth))
        assert os.path.basename(os.path.dirname(str(pth))) == f"STEP_{step}", pth
        r0, _, c0, _ = (int(v) for v in nm[:-4].split("_"))
        asked.append(nm)
        return state_tile(r0 // 256, c0 // 256, chn)

    ds.zarr.load = fake_load
    d = ds.MBADataset_tst.__new__(ds.MBADataset_tst)
    d.size, d.pad, d.chn, d.gblk, d.wid = 256, 32, chn, 16, 52 * 8
    d.gsz = (256 + 64) // 16
    d.hst, d.wst, d.hed, d.wed = hst, wst, hst + hnm, wst + wnm
    d.idir = "STEP"
    out["im/params"] = np.array([chn, hst, wst, hnm, wnm, step], dtype=np.int64)
    for (lr, c) in ((1, 1), (0, 0), (2, 2), (0, 1), (1, 2)):
        roi = [(hst + lr) * 256, (hst + lr + 1) * 256, (wst + c) * 256, (wst + c + 1) * 256]
        asked.clear()
        t, stp = d._pad_im(np.array(roi), step)
        assert stp == step and t.shape == (320, 320, chn) and t.dtype == torch.float32
        out[f"im/{lr}_{c}"] = t.numpy().astype(np.float16)                 # every value is float16-exact (or -1)
        assert np.array_equal(out[f"im/{lr}_{c}"].astype(np.float32), t.numpy())
        print(f"_pad_im tile ({lr}, {c}): {len(asked)} neighbour tiles read, {(t == -1).sum().item()} cells at -1")
    np.savez_compressed(os.path.join(GOLD, "io_ref_pad.npz"), **out)
    print("wrote", os.path.join(GOLD, "io_ref_pad.npz"), os.path.getsize(os.path.join(GOLD, "io_ref_pad.npz")), "bytes")


if __name__ == "__main__":
    main()
