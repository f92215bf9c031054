"""The configuration-surface points of SURVEY.md 8(f) row f4 (shared by the CPU and GPU tests and by
oracle/make_config_golden.py): (patch_size, rna_slc, stain, rna_num)."""
import torch

from teramind_amd import synth
from teramind_amd.config import PathConfig

CONFIGS = [(64, 1, "all", 229), (32, 4, "all", 229), (64, 8, "all", 229), (128, 4, "DAPI", 229), (64, 16, "all", 229),
           (64, 4, "all", 500), (64, 1, "all", 81), (32, 1, "PolyT", 229), (128, 1, "all", 229), (32, 16, "all", 229)]


def tag_of(c):
    return f"ps{c[0]}_z{c[1]}_{c[2]}_g{c[3]}"


def path_config(c, **kw):
    return PathConfig(patch_size=c[0], rna_slc=c[1], stain=c[2], rna_num=c[3], **kw)


def inputs(cfg):
    ne = 4
    x = synth.normal("cfg/x", (ne, cfg.in_channels, cfg.patch_size, cfg.patch_size), 0)
    rna = synth.gene_counts("cfg/rna", (ne, cfg.gn_sz, cfg.gn_sz, cfg.rna_slc * 500), 0)
    return x, rna, torch.tensor([321])


def digest(t, n=512):
    import numpy as np
    f = t.reshape(-1).double().cpu()
    idx = torch.linspace(0, f.numel() - 1, n).long()
    return np.concatenate([[f.mean().item(), f.abs().max().item()], f[idx].numpy()]).astype(np.float64)
