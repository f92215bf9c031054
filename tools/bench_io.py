#!/usr/bin/env python3
"""Row f1 measurement: tm_gene_tile_dense (COO gene tile -> dense block-summed grid) against the HBM roofline.
Algorithmic bytes per launch = 16 B per COO entry (3 x int32 coords + fp32 count) + one write of the dense
[20, 20, 26000] fp32 grid (41.6 MB zero fill; the atomics land in L2).  Prints one JSON line per nnz."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import teramind_amd  # noqa: E402,F401
from teramind_amd import _lib  # noqa: E402

PEAK_HBM_GBS = 8000.0
dev = "cuda:0"
L = _lib.lib()
for nnz in (100_000, 1_000_000, 10_000_000):
    rng = np.random.default_rng(nnz)
    crd = np.stack([rng.integers(0, 512, nnz), rng.integers(0, 512, nnz), rng.integers(0, 25000, nnz)]).astype(np.int32)
    c = torch.from_numpy(crd).to(dev)
    d = torch.from_numpy(rng.integers(1, 5, nnz).astype(np.float32)).to(dev)
    out = torch.empty((20, 20, 26000), device=dev)

    def run():
        _lib.check(L.tm_gene_tile_dense(_lib.ptr(c), _lib.ptr(d), nnz, 16, -6, -6, 20, 25000, 500, _lib.ptr(out),
                                        _lib.current_stream_ptr()))
    for _ in range(5):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    alg = 16.0 * nnz + out.numel() * 4
    print(json.dumps({"op": "tm_gene_tile_dense", "nnz": nnz, "ms": round(ms, 4), "alg_bytes": alg,
                      "roofline": {"bound": "hbm", "achieved": round(alg / ms / 1e6, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                   "frac": round(alg / ms / 1e6 / PEAK_HBM_GBS, 4)}}))
