"""Stitcher: per-step state tiles -> whole-ROI slice images.

Restates the gather of the reference's infer_brn.py:57-105 (`gen_col`: load every tile of a tile column,
reorder channels '(c s) h w -> (s c) h w', map [-1, 1] -> uint8 with `((g + 1) * 127.5).astype(uint8)` in
float16, stack the tiles of the column; `gen_mba`: join the columns) without pyvips: the mosaic is one
tensor, written with PIL as plain TIFF / JPEG (the reference's tiled pyramidal OME-TIFF needs libvips and is
out of scope).  Output index `sl = s * n_stain + c` (slice-major, stain-minor), as the reference's
per-slice directories are numbered.
"""
import os
from typing import Optional, Sequence

import numpy as np
import torch

from . import formats, tiles


def to_uint8(g: torch.Tensor) -> torch.Tensor:
    """infer_brn.py:83: `((g + 1) * 127.5).astype(np.uint8)` on the float16 tile (arithmetic in float16)."""
    return ((g.half() + 1) * 127.5).to(torch.uint8)


def reorder_slices(tile: torch.Tensor, slc: int) -> torch.Tensor:
    """'(c s) h w -> (s c) h w' with s = slc (infer_brn.py:81)."""
    cs, h, w = tile.shape[-3:]
    c = cs // slc
    return tile.reshape(*tile.shape[:-3], c, slc, h, w).transpose(-4, -3).reshape(*tile.shape[:-3], cs, h, w)


def stitch_state(state: torch.Tensor, slc: int = 50, slices: Optional[Sequence[int]] = None) -> torch.Tensor:
    """[(c s), H, W] resident state (TileSweep.local_state(), already tile-contiguous) -> uint8 [(s c), H, W]."""
    out = reorder_slices(to_uint8(state), slc)
    return out if slices is None else out[list(slices)]


def stitch_dir(step_dir, hst: int, wst: int, hnm: int, wnm: int, slc: int = 50,
               slices: Optional[Sequence[int]] = None, size: int = tiles.TILE) -> np.ndarray:
    """Mosaic of a reference-format step directory ('{r0}_{r1}_{c0}_{c1}.zip' tiles; infer_brn.py:70-76 with
    is_gen=True) -> uint8 [n_slices, hnm*size, wnm*size]."""
    out = None
    for pw in range(wnm):
        for ph in range(hnm):
            r0, c0 = hst + ph * size, wst + pw * size
            a = formats.read_state_tile(os.path.join(str(step_dir), f"{r0}_{r0 + size}_{c0}_{c0 + size}.zip"))
            t = stitch_state(torch.from_numpy(a), slc, slices).numpy()
            if out is None:
                out = np.zeros((t.shape[0], hnm * size, wnm * size), dtype=np.uint8)
            out[:, ph * size:(ph + 1) * size, pw * size:(pw + 1) * size] = t
    return out


def save_slices(mosaic, odir, names: Optional[Sequence[int]] = None, jpeg: bool = True):
    """'{odir}/all_{sl}.tif' (+ '.jpg'), the file names of infer_brn.py:96-101 (flat, non-pyramidal)."""
    from PIL import Image
    os.makedirs(str(odir), exist_ok=True)
    arr = mosaic.cpu().numpy() if isinstance(mosaic, torch.Tensor) else np.asarray(mosaic)
    for k in range(arr.shape[0]):
        sl = k if names is None else names[k]
        im = Image.fromarray(arr[k], mode="L")
        im.save(os.path.join(str(odir), f"all_{sl}.tif"))
        if jpeg:
            im.save(os.path.join(str(odir), f"all_{sl}.jpg"))
