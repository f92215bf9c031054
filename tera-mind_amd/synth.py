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
