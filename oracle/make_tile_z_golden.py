#!/usr/bin/env python3
"""Mints tests/golden/run_batch_z.json FROM THE REFERENCE (TEST INFRASTRUCTURE): the z-chunk / patchify / regroup
index maps of test_brn.Tester._run_batch (test_brn.py:183-226) for rna_slc 1, 8 and 16 -- the z_size-1 per-slice path
and the 48-slice state of the 8 / 16 configs (test_brn.py:277-278) -- captured by running the reference's own method on
a stub sampler, as oracle/make_golden.py does for rna_slc 4.  Run here:  python oracle/make_tile_z_golden.py"""
import contextlib
import importlib
import io
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_harness as rh  # noqa: E402


def digest(t, n):
    f = t.reshape(-1).double()
    idx = torch.linspace(0, f.numel() - 1, n).long()
    return {"mean": f.mean().item(), "absmax": f.abs().max().item(), "idx": idx.tolist(), "val": f[idx].tolist()}


def main():
    rh.load()
    with contextlib.redirect_stdout(io.StringIO()):
        tb = importlib.import_module("test_brn")
    out = {}
    for z, total, zpad in ((1, 50, 0), (8, 48, 1), (16, 48, 3)):
        cap, saved = {}, {}
        n_stn = 2
        cz = max(1, z // 2) * n_stn

        class StubSampler:
            def sample(self, **kw):
                cap.update({k: v for k, v in kw.items() if k != "model"})
                n = kw["shape"][0]
                return (torch.arange(n * cz * 256 * 256, dtype=torch.float32).reshape(n, cz, 256, 256) % 2039) / 64.0

        tb.zarr.save_array = lambda p, a: saved.__setitem__(str(p), a)
        t = tb.Tester.__new__(tb.Tester)
        t.gpu_id = "cpu"
        t.conf = types.SimpleNamespace(patch_size=64, gn_sz=4)
        t.z_size, t.total_slc, t.n_stn, t.epochs = z, total, n_stn, 15
        t.sampler, t.model = StubSampler(), None
        chn = total * n_stn
        gch = (50 + 2 * zpad) * 500
        tile = torch.arange(320 * 320 * chn, dtype=torch.float32).reshape(1, 320, 320, chn)
        gen = torch.Generator().manual_seed(9 + z)
        ssz = torch.Size([1, 20, 20, gch])
        crd = torch.stack([torch.randint(0, ssz[k], (4000,), generator=gen) for k in range(4)])
        crd = torch.unique(crd, dim=1)
        dat = (crd[1] * 20 * gch + crd[2] * gch + crd[3] + 1).float()
        from pathlib import Path
        t._run_batch((tile, torch.tensor([[256, 512, 512, 768]]), dat, crd, ssz, torch.tensor([3])), 3, Path("o"))
        xin, rin = cap["imgs"], cap["r_start"]
        sv = list(saved.values())[0]
        probes = [0, 7, xin.shape[0] // 2 + 3, xin.shape[0] - 1]
        out[str(z)] = {
            "total_slc": total, "gene_channels": gch, "seed": 9 + z,
            "shape": [int(v) for v in cap["shape"]], "imgs_shape": list(xin.shape), "rna_shape": list(rin.shape),
            "imgs_probe": xin[:, 1, 5, 7].long().tolist(), "imgs_sum": float(xin.double().sum()),
            "rna_nonzero": int((rin != 0).sum()), "rna_sum": float(rin.double().sum()),
            "rna_probe_rows": probes,
            "rna_probe": [[int(v) for v in rin[n].nonzero()[:3].reshape(-1).tolist()] + [float(rin[n].max())] for n in probes],
            "saved_shape": list(sv.shape), "saved_dtype": str(sv.dtype),
            "saved_digest": digest(torch.from_numpy(sv.astype(np.float32)), 97),
        }
        print("z", z, out[str(z)]["shape"], out[str(z)]["imgs_shape"], out[str(z)]["rna_shape"], out[str(z)]["saved_shape"])
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "run_batch_z.json"), "w"))


if __name__ == "__main__":
    main()
