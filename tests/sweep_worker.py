"""Rank process of the CPU multi-process sweep tests: started by teramind_amd.launch.spawn_ranks (the launcher bench.py
uses for --gpus N), runs teramind_amd.launch.run_sweep (the same per-rank entry bench.py --sweep and tools/run_roi.py run
on GPUs) under gloo with the stand-in sampler of tests/test_dist_gloo.py, and saves the rank's rows.
argv: out_dir hnm wnm state"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import teramind_amd  # noqa: E402,F401
import torch  # noqa: E402

from teramind_amd import launch  # noqa: E402
from teramind_amd.config import PathConfig  # noqa: E402
import test_dist_gloo as tdg  # noqa: E402


def main():
    out_dir, hnm, wnm, state = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    torch.set_num_threads(2)
    rank, _, world = launch.init_distributed("gloo")
    res = launch.run_sweep(PathConfig(), tdg.StandInSampler(), None, tdg.gene_provider, hnm=hnm, wnm=wnm, total_epochs=tdg.T,
                           steps=tdg.T, warmup=0, device="cpu", total_slc=tdg.SLC, hst=512, wst=768, batch_tiles=2,
                           init="reference", state=state, prefetch_genes=False)
    sw = res["sweep"]
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), sw.local_state().float().numpy())
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
        f.write(f"{sw.r0} {sw.r1} {res['world']} {res['backend']} {res['exchange_bytes_per_step']} {sw.exchanges}\n")
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
