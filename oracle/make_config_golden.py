#!/usr/bin/env python3
"""Mints tests/golden/unet_configs.npz FROM THE REFERENCE (TEST INFRASTRUCTURE): UNet outputs of the other
configuration-surface points (patch_size 32 / 128, rna_slc 1 / 8 / 16, single stains, the
500-gene mice, the 81-gene human-brain subset) for hashed weights and seeded inputs, b = 1, P = 1.
Stored per config: pred and pred2 digests (mean, |max|, 512 strided samples) -- inputs and weights are
regenerated from (tag, seed, index) by teramind_amd.synth / .weights.  Run here:  python oracle/make_config_golden.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import teramind_amd  # noqa: E402,F401
from oracle import ref_harness as rh  # noqa: E402
from teramind_amd import synth  # noqa: E402
from teramind_amd.config import PathConfig  # noqa: E402
from teramind_amd.weights import hashed_state_dict  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tests"))
from config_cases import CONFIGS, digest, inputs, tag_of  # noqa: E402


def main():
    torch.set_num_threads(8)
    out = {}
    for c in CONFIGS:
        size, srna, stain, nrna = c
        cfg = PathConfig(patch_size=size, rna_slc=srna, stain=stain, rna_num=nrna)
        model = rh.make_model(rh.make_conf(size=size, stain=stain, nrna=nrna, srna=srna))
        model.load_state_dict(hashed_state_dict(cfg, 0), strict=True)
        x, rna, t = inputs(cfg)
        with torch.inference_mode():
            o = model(x=x, t=t, rna=rna, imgs=torch.zeros(1, cfg.in_channels, size, size), patch_size=size)
        out[tag_of(c) + "/pred"] = digest(o.pred)
        out[tag_of(c) + "/pred2"] = digest(o.pred2)
        print(tag_of(c), "pred std %.4f" % o.pred.std().item(), flush=True)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "unet_configs.npz"), **out)


if __name__ == "__main__":
    main()
