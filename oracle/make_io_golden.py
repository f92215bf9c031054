#!/usr/bin/env python3
"""Mints tests/golden/io_blosc_frames.npz and tests/golden/io_state_tile_blosc.zip  (TEST INFRASTRUCTURE).

The reference writes its per-step state tiles with `zarr.save_array` (test_brn.py:225), whose default chunk
encoding in zarr 2.14.1 / numcodecs 0.15.0 (environment.yml:172,220) is Blosc-1 (lz4, clevel 5, byte
shuffle).  Neither package exists in this image, but the image's conda tree carries the C-Blosc library
itself (/opt/conda/lib/libblosc.so.1, v1.21.0 -- the same 1.21 line numcodecs 0.15 vendors).  This script
drives that library through ctypes to produce REAL Blosc frames of seeded arrays, so that the decoder in
tera-mind_amd/csrc/tm_io.hip is pinned against C-Blosc output rather than against our own encoder.
The zip fixture wraps two such frames in the zarr-v2 ZipStore layout ('.zarray' + chunk keys 'i.j.k').
Run once in this container:  python oracle/make_io_golden.py
"""
import ctypes as C
import io
import json
import os
import zipfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
LIBBLOSC = "/opt/conda/lib/libblosc.so.1"


def blosc():
    L = C.CDLL(LIBBLOSC)
    L.blosc_compress_ctx.restype = C.c_int
    L.blosc_compress_ctx.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.c_char_p, C.c_size_t, C.c_int]
    L.blosc_get_version_string.restype = C.c_char_p
    return L


def compress(L, arr, clevel=5, shuffle=1, blocksize=0, cname=b"lz4"):
    a = np.ascontiguousarray(arr)
    cap = a.nbytes + 16 + 4 * 4096
    dst = C.create_string_buffer(cap)
    n = L.blosc_compress_ctx(clevel, shuffle, a.dtype.itemsize, a.nbytes, a.ctypes.data, dst, cap, cname, blocksize, 1)
    assert n > 0, n
    return np.frombuffer(dst.raw[:n], dtype=np.uint8).copy()


def main():
    L = blosc()
    rng = np.random.default_rng(7)
    smooth16 = np.tanh(np.cumsum(rng.normal(size=300000)) / 50).astype(np.float16)       # state-like, compressible
    state16 = rng.normal(size=20000).astype(np.float16).clip(-1, 1)
    quant16 = (np.round(rng.normal(size=30000) * 4) / 4).astype(np.float16)                 # few distinct values
    counts32 = rng.poisson(0.05, size=25000).astype(np.float32)
    noise8 = rng.integers(0, 256, size=5000, dtype=np.uint8)
    cases = {
        "f16_default": (smooth16, dict()),                                    # zarr default: lz4, clevel 5, shuffle; 256 KiB blocks + leftover
        "f16_small_default": (smooth16[:30000], dict()),                      # one block
        "f16_blocks64k_leftover": (smooth16[:100001], dict(blocksize=4096)),  # forced block size (C-Blosc rounds it to 64 KiB) + short last block
        "f16_noshuffle": (quant16, dict(shuffle=0)),
        "f16_gauss": (state16, dict()),                                       # nearly incompressible splits stored raw
        "f16_clevel0_memcpy": (state16[:3000], dict(clevel=0)),               # MEMCPYED frame
        "f16_tiny_nosplit": (smooth16[:100], dict()),                         # < 128 elements: single stream
        "f32_counts": (counts32, dict()),                                     # typesize 4 -> 4 splits
        "u8_random": (noise8, dict()),                                        # typesize 1: shuffle is a no-op
        "f16_clevel9_blocks": (quant16, dict(clevel=9, blocksize=8192)),
    }
    out = {"blosc_version": np.frombuffer(L.blosc_get_version_string(), dtype=np.uint8)}
    for name, (arr, kw) in cases.items():
        fr = compress(L, arr, **kw)
        out[f"{name}/frame"] = fr
        out[f"{name}/plain"] = np.frombuffer(np.ascontiguousarray(arr).tobytes(), dtype=np.uint8)
        print(f"{name:26s} {arr.nbytes:7d} -> {fr.size:7d}  flags=0x{fr[2]:02x} typesize={fr[3]} "
              f"blocksize={int.from_bytes(fr[8:12].tobytes(), 'little')}")
    np.savez_compressed(os.path.join(GOLD, "io_blosc_frames.npz"), **out)

    # a small state tile [(stain z)=6, 40, 48] float16 chunked (3, 40, 48) the way zarr lays a ZipStore out
    tile = np.tanh(np.cumsum(rng.normal(size=(6, 40, 48)), axis=2) / 8).astype(np.float16)
    meta = {"chunks": [3, 40, 48],
            "compressor": {"blocksize": 0, "clevel": 5, "cname": "lz4", "id": "blosc", "shuffle": 1},
            "dtype": "<f2", "fill_value": 0.0, "filters": None, "order": "C", "shape": [6, 40, 48], "zarr_format": 2}
    path = os.path.join(GOLD, "io_state_tile_blosc.zip")
    with zipfile.ZipFile(path, "w", compression=zipfile.ZIP_STORED) as z:
        z.writestr(".zarray", json.dumps(meta, indent=4, sort_keys=True))
        for k in range(2):
            z.writestr(f"{k}.0.0", compress(L, tile[3 * k:3 * k + 3]).tobytes())
    np.save(os.path.join(GOLD, "io_state_tile_expected.npy"), tile)
    print("wrote", path)


if __name__ == "__main__":
    main()
