"""Deterministic synthetic inputs of the path's shapes (no dataset / checkpoint is available
offline): diffusion state / noise ~ N(0,1) and sparse non-negative integer gene counts, both
pure functions of (tag, seed, index) through the same integer hash as the weight generator."""
import numpy as np
import torch

from .weights import hashed_uniform


def normal(tag: str, shape, seed: int = 0) -> torch.Tensor:
    """Standard normal fp32 tensor (Box-Muller on two hashed uniforms)."""
    n = int(np.prod(shape))
    u1 = (hashed_uniform(tag + "/u1", n, seed) + 1.0) * 0.5
    u2 = (hashed_uniform(tag + "/u2", n, seed) + 1.0) * 0.5
    u1 = np.maximum(u1, 2.0 ** -53)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return torch.from_numpy(z.astype(np.float32).reshape(shape))


def gene_counts(tag: str, shape, seed: int = 0, density: float = 0.02, max_count: int = 3) -> torch.Tensor:
    """Dense fp32 [.., gn, gn, rna_slc*500] of block-summed-transcript-like counts: zero with
    probability 1-density, else an integer in 1..max_count."""
    n = int(np.prod(shape))
    u = (hashed_uniform(tag + "/p", n, seed) + 1.0) * 0.5
    v = (hashed_uniform(tag + "/c", n, seed) + 1.0) * 0.5
    cnt = np.where(u < density, 1.0 + np.floor(v * max_count), 0.0)
    return torch.from_numpy(cnt.astype(np.float32).reshape(shape))


def dense_to_coo(rna: torch.Tensor):
    """(dat, crd, ssz) triple as the reference's collate builds it (MBADataset_tst.py:131-148)."""
    sp = rna.to_sparse()
    return sp.values(), sp.indices(), torch.Size(rna.shape)


def write_gene_tile_dir(gdir, hnm: int, wnm: int, nnz_per_block: int, hst: int = 256, wst: int = 256, total_slc: int = 50,
                        seed: int = 0, spoil=None):
    """Synthetic gene tiles in the reference's on-disk format ('{r0}_{r1}_{c0}_{c1}_{R0}_{R1}_{C0}_{C1}.npz' COO archives of
    the tile padded by 128 px, utils/MBADataset_tst.py:65-79) CUT FROM ONE GENE MAP: the map is generated per 128 x 128 px
    block (seeded by the block's global position), a tile holds the 4 x 4 blocks of its padded ROI -- neighbouring tiles
    therefore agree where they overlap, as tiles cut from a real transcript table do.  nnz_per_block entries per block
    (16 x that per tile).  spoil = (row, col): that one tile gets one extra count in a cell it shares with its right-hand
    neighbour (a data set that violates the agreement; test hook)."""
    import os
    from . import formats
    os.makedirs(gdir, exist_ok=True)
    chan = total_slc * 500

    def block(br, bc):
        rng = np.random.default_rng([seed, br + 4096, bc + 4096])
        return (rng.integers(0, 128, nnz_per_block), rng.integers(0, 128, nnz_per_block), rng.integers(0, chan, nnz_per_block),
                rng.integers(1, 4, nnz_per_block).astype(np.uint16))

    for r in range(hst // 256, hst // 256 + hnm):
        for c in range(wst // 256, wst // 256 + wnm):
            ys, xs, cs, ds = [], [], [], []
            for i in range(4):
                for j in range(4):
                    y, x, ch, d = block(2 * r - 1 + i, 2 * c - 1 + j)
                    ys.append(y + 128 * i); xs.append(x + 128 * j); cs.append(ch); ds.append(d)
            crd = np.stack([np.concatenate(ys), np.concatenate(xs), np.concatenate(cs)]).astype(np.int64)
            data = np.concatenate(ds)
            if spoil is not None and (r, c) == tuple(spoil):
                crd = np.concatenate([crd, np.array([[200], [400], [7]], dtype=np.int64)], axis=1)      # px (200, 400): shared with col + 1
                data = np.concatenate([data, np.array([1], dtype=np.uint16)])
            v = (r * 256, r * 256 + 256, c * 256, c * 256 + 256, r * 256 - 128, r * 256 + 384, c * 256 - 128, c * 256 + 384)
            formats.write_gene_npz(os.path.join(gdir, "_".join(map(str, v)) + ".npz"), data, crd, (512, 512, chan))
