"""Gene-gene attention read-out of the attention driver (reference test_attn.py:359-431, `--calc_attn`).

`run_attn_batch` is `Tester._run_batch` of test_attn re-stated over the HIP attention-map model:
z-chunk the gene tile into 25 windows of 4 slices, cut the 20x20 gene grid into 5x5 patches of 4x4
cells, get the four softmax maps per patch (tm_gene_attn), contract the maps of the selected pathway
genes `glst` with their counts, reassemble per tile and crop the half-patch frame.
The contractions are 2x2 / 4x2 matrices times 2x16 count blocks per patch -- host glue on device
tensors, not a kernel.
"""
from typing import Sequence

import torch

from . import tiles

PATHWAYS = {"GLUT": (75, 191), "DOPA": (5, 154), "BLOD": (94, 145)}      # gene indices, SURVEY.md section 2


def pathway_readout(attn: torch.Tensor, rna_mid: torch.Tensor, glst: Sequence[int]) -> torch.Tensor:
    """attn [4, B, G, G] (3 slice-pair maps + ensemble), rna_mid [B, G, 2, gh, gw] (middle slices) ->
    [B, 4 + 2 + 2, 2*gh*gw]: slice-pair products, ensemble product, raw counts (test_attn.py:404-423)."""
    g = list(glst)
    B = attn.shape[1]
    sub = attn[:, :, g][..., g]                                  # [4, B, 2, 2]
    rna = rna_mid[:, g]                                          # [B, 2, 2, gh, gw]
    rna0, rna1 = rna[:, :, 0].reshape(B, len(g), -1), rna[:, :, 1].reshape(B, len(g), -1)
    att0 = sub[:2].permute(1, 0, 2, 3).reshape(B, 2 * len(g), len(g))       # 'd b c g -> b (d c) g'
    att1 = sub[1:3].permute(1, 0, 2, 3).reshape(B, 2 * len(g), len(g))
    out = torch.cat([att0 @ rna0, att1 @ rna1], -1)              # [B, 4, 2*gh*gw]
    rna2 = rna.reshape(B, len(g), -1)
    return torch.cat([out, sub[3] @ rna2, rna2], 1)


def assemble_readout(out: torch.Tensor, b: int, p1: int, p2: int, gn: int) -> torch.Tensor:
    """'(n_z b p1 p2) g (z h w) -> b (n_z z) g (p1 h) (p2 w)' then crop gn//2 cells (test_attn.py:424-427)."""
    n, g, zhw = out.shape
    z = zhw // (gn * gn)
    nz = n // (b * p1 * p2)
    t = out.reshape(nz, b, p1, p2, g, z, gn, gn).permute(1, 0, 5, 4, 2, 6, 3, 7)
    t = t.reshape(b, nz * z, g, p1 * gn, p2 * gn)
    pad = gn // 2
    return t[:, :, :, pad:-pad, pad:-pad]


def run_attn_batch(model, rna_tile: torch.Tensor, glst: Sequence[int], z_size: int = 4, gn: int = 4) -> torch.Tensor:
    """rna_tile dense [b, 20, 20, (50+2)*500] -> fp16 [b, 50, 8, 16, 16] (what the reference saves per tile)."""
    b = rna_tile.shape[0]
    rna = tiles.zchunk_rna(rna_tile, z_size)
    p1, p2 = rna.shape[1] // gn, rna.shape[2] // gn
    rna = tiles.patchify_hwc(rna, gn, False)
    attn, mid = model.forward(x=None, t=None, rna=rna, imgs=None)
    out = pathway_readout(attn, mid, glst)
    return assemble_readout(out, b, p1, p2, gn).half()
