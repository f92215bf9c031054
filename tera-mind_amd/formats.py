"""On-disk formats either side of the denoising path.

The reference moves every tile through the filesystem each diffusion step:
  * gene tile  = `sparse.save_npz` COO archive named '{r0}_{r1}_{c0}_{c1}_{R0}_{R1}_{C0}_{C1}.npz'
    (inner ROI + ROI padded by 128 px; test_brn.py:51-70, utils/MBADataset_tst.py:65-79), shape
    [512, 512, 50*500];
  * state tile = `zarr.save_array('{r0}_{r1}_{c0}_{c1}.zip', out)` (test_brn.py:222-226): a zarr-v2 array
    in a ZipStore, float16 [100, 256, 256], channel order (stain, z); read back with `zarr.load`
    (utils/MBADataset_tst.py:60, infer_brn.py:76).
This build keeps the state resident (brain.TileSweep) and only touches these formats at the edges:
reading gene tiles, importing / exporting a step directory for interoperability and resume.

Neither `zarr`, `numcodecs` nor `sparse` exist in this image, so both containers are restated from
their published layouts (zarr storage spec v2; pydata/sparse 0.15.5 `save_npz`; c-blosc 1.x frames):
  * readers accept what zarr 2.14.1 writes by default (Blosc lz4 + byte shuffle, decoded by
    `tm_blosc_decompress` in the C-ABI library), zlib, or no compressor;
  * writers emit spec-conformant archives that `zarr.load` / `sparse.load_npz` open
    (state tiles: one uncompressed or zlib chunk).
Only safe loaders are used (`numpy.load(allow_pickle=False)`, `zipfile`, `json`).
"""
import ctypes as C
import io
import json
import os
import zipfile
import zlib
from itertools import product
from typing import Optional, Sequence, Tuple

import numpy as np

GENES_PER_SLICE = 500


# ---- names ------------------------------------------------------------------------------------
def parse_gene_tile_name(path) -> Tuple[Tuple[int, ...], Tuple[int, ...]]:
    """'{r0}_{r1}_{c0}_{c1}_{R0}_{R1}_{C0}_{C1}.npz' -> (roi, roio)  (utils/MBADataset_tst.py:147-150)."""
    stem = os.path.splitext(os.path.basename(str(path)))[0]
    v = [int(p) for p in stem.split("_")]
    if len(v) != 8:
        raise ValueError(f"gene tile name needs 8 integers, got {stem!r}")
    return tuple(v[:4]), tuple(v[4:])


def parse_state_tile_name(path) -> Tuple[int, int, int, int]:
    stem = os.path.splitext(os.path.basename(str(path)))[0]
    v = [int(p) for p in stem.split("_")]
    if len(v) != 4:
        raise ValueError(f"state tile name needs 4 integers, got {stem!r}")
    return tuple(v)


# ---- gene tiles: pydata/sparse COO .npz ----------------------------------------------------------
def read_gene_npz(path):
    """-> (data [nnz], coords int64 [ndim, nnz], shape tuple).  Layout of sparse.save_npz (0.15.5):
    arrays 'data', 'coords', 'shape', 'fill_value' in a (compressed) .npz."""
    with np.load(path, allow_pickle=False) as z:
        data, coords, shape = z["data"], z["coords"], tuple(int(s) for s in z["shape"])
        if "fill_value" in z.files and float(z["fill_value"][()]) != 0.0:
            raise ValueError("gene tile with a non-zero fill value")
    if coords.ndim != 2 or coords.shape[0] != len(shape) or coords.shape[1] != data.shape[0]:
        raise ValueError(f"malformed COO archive {path}: coords {coords.shape}, data {data.shape}, shape {shape}")
    return data, coords.astype(np.int64, copy=False), shape


def write_gene_npz(path, data, coords, shape, compressed: bool = True):
    coords = np.asarray(coords)
    save = np.savez_compressed if compressed else np.savez
    with open(path, "wb") as f:                       # file object: numpy must not append '.npz'
        save(f, data=np.asarray(data), coords=coords, shape=np.asarray(shape, dtype=np.int64),
             fill_value=np.asarray(np.zeros((), dtype=np.asarray(data).dtype)))


def gene_tile_shift(roi: Sequence[int], roio: Sequence[int], gblk: int = 16, pad: int = 32) -> Tuple[int, int]:
    """Cell shift `psz - (roi[2i] - roio[2i]) // gblk` of _pad_gn (utils/MBADataset_tst.py:84-86)."""
    psz = pad // gblk
    return psz - (roi[0] - roio[0]) // gblk, psz - (roi[2] - roio[2]) // gblk


# ---- state tiles: zarr v2 array in a zip ------------------------------------------------------------
def _blosc_decode(buf: bytes) -> bytes:
    from . import _lib
    L = _lib.lib()
    n = C.c_size_t(0)
    src = (C.c_char * len(buf)).from_buffer_copy(buf)
    _lib.check(L.tm_blosc_decompress(src, len(buf), None, 0, C.byref(n)), "tm_blosc_decompress(size)")
    out = (C.c_char * max(1, n.value))()
    _lib.check(L.tm_blosc_decompress(src, len(buf), out, n.value, C.byref(n)), "tm_blosc_decompress")
    return bytes(out[:n.value])


def _decode_chunk(buf: bytes, compressor: Optional[dict]) -> bytes:
    if compressor is None:
        return buf
    cid = compressor.get("id")
    if cid == "blosc":
        return _blosc_decode(buf)
    if cid in ("zlib", "gzip"):
        return zlib.decompress(buf, 15 + 32)
    raise NotImplementedError(f"zarr compressor {cid!r} (supported: blosc-lz4, zlib, gzip, none)")


def read_zarr_zip(path) -> np.ndarray:
    """What `zarr.load(path)` returns for a single-array zarr-v2 ZipStore."""
    with zipfile.ZipFile(path) as z:
        names = set(z.namelist())
        if ".zarray" not in names:
            raise ValueError(f"{path}: no .zarray at the archive root (not a zarr-v2 array store)")
        meta = json.loads(z.read(".zarray"))
        if meta.get("zarr_format") != 2:
            raise ValueError(f"{path}: zarr_format {meta.get('zarr_format')!r}, expected 2")
        if meta.get("filters"):
            raise NotImplementedError("zarr filters")
        shape, chunks = tuple(meta["shape"]), tuple(meta["chunks"])
        dtype, order = np.dtype(meta["dtype"]), meta.get("order", "C")
        sep = meta.get("dimension_separator", ".")
        fill = meta.get("fill_value")
        out = np.empty(shape, dtype=dtype)
        out[...] = 0 if fill is None else (np.nan if fill == "NaN" else fill)
        grid = [range((s + c - 1) // c) for s, c in zip(shape, chunks)]
        for idx in product(*grid):
            key = sep.join(str(i) for i in idx) if idx else "0"
            if key not in names:
                continue                                  # absent chunk = fill value
            raw = _decode_chunk(z.read(key), meta.get("compressor"))
            blk = np.frombuffer(raw, dtype=dtype, count=int(np.prod(chunks))).reshape(chunks, order=order)
            sl = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, shape))
            out[sl] = blk[tuple(slice(0, s.stop - s.start) for s in sl)]       # edge chunks are stored full size
    return out


def write_zarr_zip(path, arr: np.ndarray, compressor: Optional[str] = None, level: int = 1):
    """A single-chunk zarr-v2 array in a ZIP_STORED ZipStore (the layout zarr.save_array produces, with the
    chunk either raw or zlib-compressed so that no Blosc encoder is needed)."""
    arr = np.ascontiguousarray(arr)
    if compressor not in (None, "zlib"):
        raise NotImplementedError("writer supports compressor None or 'zlib'")
    meta = {"chunks": list(arr.shape), "compressor": None if compressor is None else {"id": "zlib", "level": level},
            "dtype": arr.dtype.str, "fill_value": 0.0 if arr.dtype.kind == "f" else 0, "filters": None, "order": "C",
            "shape": list(arr.shape), "zarr_format": 2}
    payload = arr.tobytes()
    if compressor == "zlib":
        payload = zlib.compress(payload, level)
    tmp = str(path) + ".tmp"
    with zipfile.ZipFile(tmp, "w", compression=zipfile.ZIP_STORED, allowZip64=True) as z:
        z.writestr(".zarray", json.dumps(meta, indent=4, sort_keys=True))
        z.writestr(".".join("0" for _ in arr.shape) if arr.ndim else "0", payload)
    os.replace(tmp, path)                                  # a reader never sees a half-written tile


def read_state_tile(path) -> np.ndarray:
    """float16 [(stain z), 256, 256] as written by test_brn.py:222-226."""
    a = read_zarr_zip(path)
    if a.ndim != 3:
        raise ValueError(f"{path}: state tile must be 3-D, got {a.shape}")
    return a


def write_state_tile(path, tile, compressor: Optional[str] = None):
    write_zarr_zip(path, np.asarray(tile, dtype=np.float16), compressor)
