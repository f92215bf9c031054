"""Inputs of the training-objective tests, shared with oracle/make_train_golden.py:
b = 2 images of 2 x 2 patches, the reference's own padding / grid / mask construction (experiment.py:155-168)."""
import torch
import torch.nn.functional as F

from teramind_amd import synth

CASES = {"mse_seed3": (3, "mse"), "l1_seed8": (8, "l1")}


def make_inputs(seed, b=2, ps=64, n=2, C=4, srna=4):
    imgs = synth.normal(f"train/img{seed}", (b, C, n * ps, n * ps), seed).clamp(-1, 1)
    halfp = ps // 2
    x_pad = F.pad(imgs, (halfp, halfp, halfp, halfp))
    g = torch.linspace(0, n, n + 1)
    xx, yy = torch.meshgrid(g, g, indexing="ij")
    pos = torch.stack([xx, yy], dim=-1)
    mask = torch.zeros_like(x_pad)
    mask[:, :, halfp:-halfp, halfp:-halfp] = 1.0
    cells = x_pad.shape[2] // 16
    dense = synth.gene_counts(f"train/rna{seed}", (b, cells, cells, srna * 500), seed)
    crd = dense.nonzero().t().contiguous()
    dat = dense[tuple(crd)]
    t = torch.tensor([(211 * (i + 1) + 97 * seed) % 1000 for i in range(b)], dtype=torch.long)
    noise = synth.normal(f"train/noise{seed}", tuple(x_pad.shape), seed + 1)
    idx = torch.arange(b)
    return x_pad, (dat, crd, torch.Size([b, cells, cells, srna * 500])), imgs, t, pos, mask, idx, noise
