#!/usr/bin/env python3
"""Mints tests/golden/train_attn_ref.npz FROM THE REFERENCE (TEST INFRASTRUCTURE): the reference's own `AttnBlock`
(model/MBAblocks.py:428-514, gene_trans=True, num_heads=1, z_size=2, n_h=2 -- the configuration model/MBAModel.py builds
for every gene cross-attention block) run forward and backward through torch.autograd on CPU, on the seeded inputs and
parameters of tests/train_cases.py.  Stored: out, dL/dx, dL/dcond and the gradient of every parameter for L = sum(out * dout).
The one restated piece inside is timm's `Mlp` (oracle/ref_harness.py header).  Run here:  python oracle/make_train_block_golden.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import teramind_amd  # noqa: E402,F401
from oracle import ref_harness as rh  # noqa: E402
from train_cases import ATTN_CASES, make_attn_inputs  # noqa: E402


def main():
    torch.set_num_threads(8)
    rh.load()
    from model.MBAblocks import AttnBlock
    out = {}
    for name, c in ATTN_CASES.items():
        x, cond, dout, params = make_attn_inputs(name)
        blk = AttnBlock(c["C"], num_heads=1, enable_flash_attn=True, gene_trans=True, gene_size=c["G"], z_size=2, n_h=2).double()
        blk.load_state_dict({k: v.double() for k, v in params.items()}, strict=True)
        blk.train()
        xx, cc = x.double().requires_grad_(True), cond.double().requires_grad_(True)
        y = blk(xx, None, cc)
        (y * dout.double()).sum().backward()
        out[f"{name}/out"] = y.detach().float().numpy()
        out[f"{name}/dx"] = xx.grad.float().numpy()
        out[f"{name}/dcond"] = cc.grad.float().numpy()
        for k, p in blk.named_parameters():
            out[f"{name}/grad/{k}"] = p.grad.float().numpy()
        print(name, "out", float(y.abs().mean()), "dx", float(xx.grad.abs().mean()), "dcond", float(cc.grad.abs().mean()), flush=True)
    p = os.path.join(ROOT, "tests", "golden", "train_attn_ref.npz")
    np.savez_compressed(p, **out)
    print("wrote", p, os.path.getsize(p), "bytes")


if __name__ == "__main__":
    main()
