"""Index / layout logic of the tiled whole-brain sampler (host side, device agnostic).

Restates the integer and layout paths of the reference inference driver:
  * `lcg`, per-tile noise seeds and the step-0 noise tile  (utils/MBADataset_tst.py:13-14,49-63)
  * gene-tile file names / ROI arithmetic                     (test_brn.py:51-70)
  * z-chunking, patchify, regroup of Tester._run_batch        (test_brn.py:183-208,219-221)
  * the contiguous row-block partition of the tile grid that replaces DistributedSampler
    (test_brn.py:44).
All functions are pure torch / Python and run on CPU or GPU tensors alike (plumbing only;
no arithmetic of the model lives here).
"""
from typing import List, Tuple

import torch

TILE = 256                 # size of one output tile (test_brn.py:51 `size=256`)
NOISE_GRID_WIDTH = 52 * 8  # `wid` of MBADataset_tst (utils/MBADataset_tst.py:24)
GENES = 500


def lcg(x: int, a: int = 1103515245, c: int = 12345, m: int = 2 ** 31) -> int:
    return (a * x + c) % m


def tile_noise_seed(row: int, col: int, wid: int = NOISE_GRID_WIDTH) -> int:
    """Seed of the step-0 noise of the tile at absolute grid position (row, col)."""
    return lcg(row * wid + col)


def initial_noise_tile(row: int, col: int, chn: int, size: int = TILE) -> torch.Tensor:
    """The reference's step-0 tile: CPU mt19937 stream, shape (size, size, chn) 'h w c'.
    (torch.manual_seed is global state in the reference too; the generator here is local.)"""
    g = torch.Generator(device="cpu")
    g.manual_seed(tile_noise_seed(row, col))
    return torch.randn((size, size, chn), generator=g)


def gene_tile_names(size: int = TILE, hst: int = 256, wst: int = 256, hnm: int = 286, wnm: int = 414) -> List[str]:
    """'{r0}_{r1}_{c0}_{c1}_{R0}_{R1}_{C0}_{C1}.npz' stems, column-major like the reference list."""
    pad, names = size // 2, []
    for pw in range(wnm):
        for ph in range(hnm):
            r0, c0 = hst + ph * size, wst + pw * size
            vals = (r0, r0 + size, c0, c0 + size, r0 - pad, r0 + size + pad, c0 - pad, c0 + size + pad)
            names.append("_".join(str(v) for v in vals))
    return names


def state_tile_name(row: int, col: int, size: int = TILE) -> str:
    """'{r0}_{r1}_{c0}_{c1}' of a saved state tile (test_brn.py:225, MBADataset_tst.py:103-104)."""
    return f"{row * size}_{(row + 1) * size}_{col * size}_{(col + 1) * size}"


def state_slices(rna_slc: int) -> int:
    """Brain slices held in a state tile: 48 for the 8- and 16-slice models (their z windows of 8 / 16 gene slices,
    stride 4 / 8, centre on 4 / 8 state slices and the outermost gene slices have no state), 50 otherwise
    (test_brn.py:277-278 `_chn`)."""
    return 48 if rna_slc in (8, 16) else 50


def row_block_partition(hnm: int, world: int) -> List[Tuple[int, int]]:
    """[r0, r1) tile rows per rank: contiguous blocks, sizes differ by at most one row."""
    base, extra = divmod(hnm, world)
    out, r = [], 0
    for k in range(world):
        n = base + (1 if k < extra else 0)
        out.append((r, r + n))
        r += n
    return out


# ---- Tester._run_batch layout maps ---------------------------------------------------------
def zchunk_state(tile: torch.Tensor, total_slc: int, z_size: int) -> torch.Tensor:
    """[b, h, w, (s z)] with z = total_slc  ->  [(n_z b), h, w, (s zc)], zc = z_size // 2
    (test_brn.py:188-192); z_size 1: one slice per model call, '(z b) h w s' (test_brn.py:183-185)."""
    b, h, w, sz = tile.shape
    s, zc = sz // total_slc, max(1, z_size // 2)
    nz = total_slc // zc
    return tile.reshape(b, h, w, s, nz, zc).permute(4, 0, 1, 2, 3, 5).reshape(nz * b, h, w, s * zc)


def zchunk_rna(rna: torch.Tensor, z_size: int) -> torch.Tensor:
    """[b, h, w, (z g)] -> windows of z_size slices, stride z_size//2 -> [(n_s b), h, w, (s g)]
    (test_brn.py:193-197); z_size 1: '(z b) h w g' (test_brn.py:186-187)."""
    b, h, w, zg = rna.shape
    win = rna.reshape(b, h, w, zg // GENES, GENES).unfold(3, z_size, max(1, z_size // 2))      # b h w n_s g s
    ns = win.shape[3]
    return win.permute(3, 0, 1, 2, 5, 4).reshape(ns * b, h, w, z_size * GENES)


def patchify_hwc(t: torch.Tensor, ps: int, channels_first: bool) -> torch.Tensor:
    """'b (p1 h) (p2 w) c -> (b p1 p2) c h w' (state) or '... -> (b p1 p2) h w c' (genes)."""
    b, H, W, c = t.shape
    p1, p2 = H // ps, W // ps
    t = t.reshape(b, p1, ps, p2, ps, c)
    t = t.permute(0, 1, 3, 5, 2, 4) if channels_first else t.permute(0, 1, 3, 2, 4, 5)
    return t.reshape(b * p1 * p2, *t.shape[3:]).contiguous()


def regroup_output(out: torch.Tensor, b: int, n_stain: int) -> torch.Tensor:
    """'(n_z b) (s z) h w -> b (s n_z z) h w'."""
    nzb, sz, h, w = out.shape
    nz, zc = nzb // b, sz // n_stain
    return out.reshape(nz, b, n_stain, zc, h, w).permute(1, 2, 0, 3, 4, 5).reshape(b, n_stain * nz * zc, h, w)


def run_batch_inputs(tile_hwc: torch.Tensor, rna_hwc: torch.Tensor, patch_size: int, gn_sz: int, total_slc: int,
                     z_size: int):
    """The tensors Tester._run_batch hands to sampler.sample for rna_slc in (1, 4, 8, 16):
    returns (x_patches [(n_z b p1 p2), C, ps, ps], rna_patches [(n_s b p1 p2), gn, gn, z_size*500],
    shape (n_z*b, C, H-ps, W-ps))."""
    x = zchunk_state(tile_hwc, total_slc, z_size)
    r = zchunk_rna(rna_hwc, z_size)
    shape = (x.shape[0], x.shape[3], x.shape[1] - patch_size, x.shape[2] - patch_size)
    return patchify_hwc(x, patch_size, True), patchify_hwc(r, gn_sz, False), shape
