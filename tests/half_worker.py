"""Child process of test_gpu_unet.py::test_half_resolution_conditioning_is_bit_identical: one 16-bit forward (collage
decoder, P = 2) with whatever TM_ATTN_HALF the parent put into the environment; writes pred / pred2 to argv[2]."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch            # noqa: E402
import util             # noqa: E402
from teramind_amd import synth                      # noqa: E402
from teramind_amd.config import PathConfig          # noqa: E402
from teramind_amd.unet import BeatGANsUNetModel     # noqa: E402


def run(dtype):
    dev = "cuda:0"
    cfg = PathConfig(compute_dtype=dtype)
    m = BeatGANsUNetModel(cfg, dev).load_state_dict(util.state_dict(cfg))
    b, P = 2, 2
    ne = b * (P + 1) * (P + 1)
    x = synth.normal("x", (ne, 4, 64, 64), 3)
    rna = synth.gene_counts("rna", (ne, 4, 4, 2000), 3)
    t = torch.tensor([311, 702], dtype=torch.long)
    out = m(x=x.to(dev), t=t.to(dev), rna=rna.to(dev), imgs=torch.zeros(b, 4, 64 * P, 64 * P), patch_size=64, want_pred2=True)
    torch.cuda.synchronize()
    return out.pred.cpu(), out.pred2.cpu()


if __name__ == "__main__":
    pred, pred2 = run(sys.argv[1])
    torch.save({"pred": pred, "pred2": pred2}, sys.argv[2])
