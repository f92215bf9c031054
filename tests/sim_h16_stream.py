"""Design aid (not a test): what does a 16-bit RESIDUAL STREAM cost in accuracy?  The CPU oracle is run with
(a) 16-bit conv / Linear operands only (the round-1 16-bit path: fp32 residual stream) and (b) additionally every block
output (stem, ResBlock, AttnBlock, RNA pyramid level) rounded to the 16-bit type -- what storing the inter-block
activations in bf16 / fp16 does.  Prints relative L2 / max-abs of eps against the fp32 oracle.
usage: python tests/sim_h16_stream.py [b P]"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import teramind_amd  # noqa: E402,F401
import util  # noqa: E402
from oracle import teramind_cpu as tc  # noqa: E402
from teramind_amd import synth  # noqa: E402
from teramind_amd.config import PathConfig  # noqa: E402


def run(dt, stream, x, t, rna, p, sd, oc):
    r = (lambda a: a) if dt is None else (lambda a: a.to(dt).float())
    conv3d, linear = F.conv3d, F.linear
    res_block, attn_block, rna_pyramid = tc.res_block, tc.attn_block, tc.rna_pyramid
    F.conv3d = lambda i, w, b=None, **kw: conv3d(r(i), r(w), b, **kw)
    F.linear = lambda i, w, b=None: linear(r(i), r(w), b)
    if stream:
        tc.res_block = lambda *a, **kw: r(res_block(*a, **kw))
        tc.attn_block = lambda *a, **kw: r(attn_block(*a, **kw))
        tc.rna_pyramid = lambda *a, **kw: [r(v) for v in rna_pyramid(*a, **kw)]
    try:
        with torch.inference_mode():
            return tc.unet_forward(sd, oc, x, t, rna, p, p)[0]
    finally:
        F.conv3d, F.linear = conv3d, linear
        tc.res_block, tc.attn_block, tc.rna_pyramid = res_block, attn_block, rna_pyramid


def main():
    b, P = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1, 1)
    cfg = PathConfig()
    sd, oc = util.state_dict(cfg), tc.oracle_config_from(cfg)
    p = P + 1
    for seed in (0, 3):
        x = synth.normal("x", (b * p * p, 4, 64, 64), seed)
        rna = synth.gene_counts("rna", (b * p * p, 4, 4, 2000), seed)
        t = torch.tensor([(137 * (i + 1) + 61 * seed) % 1000 for i in range(b)], dtype=torch.long)
        ref = run(None, False, x, t, rna, p, sd, oc)
        for name, dt in (("bf16", torch.bfloat16), ("f16", torch.float16)):
            for stream in (False, True):
                got = run(dt, stream, x, t, rna, p, sd, oc)
                rel = ((got - ref).norm() / ref.norm()).item()
                print(f"seed {seed} {name} stream16={int(stream)}: rel_l2 {rel:.3e} max|d| {(got - ref).abs().max().item():.3e}", flush=True)


if __name__ == "__main__":
    main()
