"""TEST INFRASTRUCTURE ONLY -- mints tests/golden/unet_autocast_ref.npz: outputs of the REAL reference UNet
(model/unet_ours.py BeatGANsUNetModel, imported from /root/reference through oracle/ref_harness.py) run under
`torch.autocast('cpu', dtype=bfloat16 | float16)` on the inputs / hashed weights of tests/golden/unet_full.npz.

Why: the reference samples under `autocast('cuda', enabled=conf.fp16)` (diffusion/base.py:377), which is inert on the CPU
where every other fixture is minted, so the 16-bit modes of the HIP path (TM_DTYPE_BF16 / TM_DTYPE_F16) had no
reference-side 16-bit vector at all and were bounded against the fp32 oracle only.  CPU autocast is the closest thing this
container can run: it applies the same cast policy to the ops that carry the model's arithmetic -- conv3d, linear, matmul /
bmm and scaled_dot_product_attention run in the 16-bit type, with fp32 accumulation inside the op and a 16-bit result --
and leaves normalisation statistics, softmax inputs of non-fused paths and elementwise glue in the dtype they arrive in,
as CUDA autocast does.  It is NOT the CUDA autocast run: the two op lists differ at the edges (e.g. CPU autocast computes
avg_pool3d in fp32), oneDNN's accumulation order is not cuDNN's, and the nearest-upsample / elementwise kernels round at the
same places but are different code.  The fixture therefore pins "the HIP 16-bit modes round where the reference's own
autocast rounds" to a tolerance (tests/test_gpu_golden.py), not bit for bit; DESIGN.md section 2 states this.

  python oracle/make_autocast_golden.py        (in the build container; ~1 min)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import teramind_amd  # noqa: F401,E402
from teramind_amd import synth  # noqa: E402
from teramind_amd.config import PathConfig  # noqa: E402
from teramind_amd.weights import hashed_state_dict  # noqa: E402
import ref_harness as rh  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
CASES = [(1, 1, 0), (1, 2, 3)]                      # (b, P, seed): the cases of unet_full.npz


def inputs(b, P, seed):
    p = P + 1
    ne = b * p * p
    x = synth.normal("x", (ne, 4, 64, 64), seed)
    rna = synth.gene_counts("rna", (ne, 4, 4, 2000), seed)
    t = torch.tensor([(137 * (i + 1) + 61 * seed) % 1000 for i in range(b)], dtype=torch.long)
    return x, t, rna


def main():
    rh.load()
    torch.manual_seed(0)
    cfg = PathConfig()
    model = rh.make_model(rh.make_conf())
    model.load_state_dict(hashed_state_dict(cfg, 0), strict=True)
    out = {}
    full = np.load(os.path.join(OUT, "unet_full.npz"))
    for b, P, seed in CASES:
        x, t, rna = inputs(b, P, seed)
        kw = dict(x=x, t=t, rna=rna, imgs=torch.zeros(b, 4, 64 * P, 64 * P), patch_size=64)
        tag = f"b{b}_P{P}_s{seed}"
        with torch.inference_mode():
            ref32 = model(**kw)
            # the harness reproduces the committed fp32 fixture (same inputs, same weights) before anything is minted from it
            assert np.array_equal(ref32.pred.numpy(), full[f"{tag}/pred"]), "fp32 run does not reproduce unet_full.npz"
            for name, dt in (("bf16", torch.bfloat16), ("f16", torch.float16)):
                with torch.autocast("cpu", dtype=dt):
                    o = model(**kw)
                pred = o.pred.float()
                rel = ((pred - ref32.pred).pow(2).mean().sqrt() / ref32.pred.pow(2).mean().sqrt()).item()
                out[f"{name}/{tag}/pred"] = pred.numpy()
                out[f"{name}/{tag}/rel_l2_vs_fp32"] = np.float64(rel)
                if P == 1:
                    out[f"{name}/{tag}/pred2"] = o.pred2.float().numpy()
                print(f"{name} {tag}: reference under CPU autocast vs reference fp32: rel L2 {rel:.3e}")
    np.savez_compressed(os.path.join(OUT, "unet_autocast_ref.npz"), **out)
    print("written", os.path.join(OUT, "unet_autocast_ref.npz"))


if __name__ == "__main__":
    main()
