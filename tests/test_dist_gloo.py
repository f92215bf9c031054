"""CPU, gloo, world_size 2 (and 3): the row-sharded tile sweep with per-step halo-strip exchange
produces bit-identical state to the single-rank sweep.  The compute is a cheap stand-in model
whose output depends on the 32-px halo (so a wrong / missing exchange changes the result) and
the oracle's CPU sampler step; the distributed / indexing logic under test is the product code
(teramind_amd.brain.TileSweep)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from oracle import teramind_cpu as tc
from teramind_amd.brain import TileSweep
from teramind_amd.config import PathConfig

T, HNM, WNM, SLC = 3, 3, 2, 4          # 3x2 tiles, 4 brain slices (2 z-chunks), 3 DDIM steps


class StandInSampler:
    """sampler.sample(...)-shaped: eps = blur of the padded 320x320 window (reads 4 px of halo)
    plus a gene term, then the oracle's DDIM step (mode B)."""

    def __init__(self):
        self.sch = tc.make_schedule(T, "ddim")

    def sample(self, model=None, shape=None, imgs=None, noise=None, r_start=None, patch_size=64, idx=None,
               model_kwargs=None, **kw):
        n, c, H, W = shape
        P1, P2 = H // patch_size, W // patch_size
        full = tc.unpatchify(imgs, P1 + 1, P2 + 1)
        hp = patch_size // 2
        eps_img = F.avg_pool2d(full, 9, stride=1, padding=4)[..., hp:-hp, hp:-hp]
        gene = r_start.reshape(n, (P1 + 1) * (P2 + 1), -1).sum((1, 2)).reshape(n, 1, 1, 1)
        eps = tc.patchify(0.3 * eps_img + 2e-5 * gene, patch_size)
        return tc.sampler_step(self.sch, "ddim", imgs, eps, idx, P1, P2)


def gene_provider(row, col):
    g = torch.Generator().manual_seed(row * 1000 + col)
    return (torch.rand((20, 20, (SLC + 2) * 500), generator=g) < 0.01).float()


def make_sweep(rank, world, state="fp32x2", batch_tiles=2, share=False):
    genes = gene_provider
    if share:                 # shared-halo windows need gene tiles that agree where they overlap
        from teramind_amd.brain import consistent_gene_provider
        genes = consistent_gene_provider(PathConfig(), "cpu", total_slc=SLC, density=0.01)
    return TileSweep(PathConfig(), StandInSampler(), None, genes, hst=512, wst=768, hnm=HNM, wnm=WNM,
                     total_epochs=T, total_slc=SLC, device="cpu", rank=rank, world=world, batch_tiles=batch_tiles,
                     state=state, share_halo=share)


def _worker(rank, world, port, q, state="fp32x2", share=False):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sw = make_sweep(rank, world, state, share=share)
        out = sw.test().float().clone()
        q.put((rank, sw.r0, sw.r1, out.numpy()))     # by value: the worker may exit before the parent reads
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,state,share", [(2, "fp32x2", False), (3, "fp32x2", False), (2, "fp16", False), (2, "fp16", True)])
def test_row_sharded_sweep_equals_single_rank(world, state, share):
    """share: one-row shared-halo windows (TileSweep(share_halo=True)) -- the same windows whatever the row partition, so
    the row-sharded result still equals the single-rank one (the stand-in model is not patch-local, hence no comparison
    with per-tile calls here: that equality is tests/test_gpu_sweep.py's, on the real model)."""
    torch.set_num_threads(4)
    ref = make_sweep(0, 1, share=share).test()
    assert ref.shape == (SLC * 2, HNM * 256, WNM * 256)
    assert float(ref.abs().max()) <= 1.0 + 1e-6 and float(ref.std()) > 0.05
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, state, share)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, r0, r1, out in got:
        assert torch.equal(torch.from_numpy(out), ref[:, r0 * 256:r1 * 256, :]), f"rank {rank} rows [{r0},{r1}) differ from the single-rank sweep"



@pytest.mark.parametrize("batch_tiles", [1, 3, 6])
def test_single_fp16_canvas_state_is_bit_identical(batch_tiles):
    """state='fp16' (one float16 canvas, rows committed one row late, step-0 noise regenerated in a band):
    the whole-brain memory layout gives exactly the two-canvas result, also with batches that span tile rows."""
    torch.set_num_threads(4)
    ref = make_sweep(0, 1).test()
    sw = make_sweep(0, 1, "fp16", batch_tiles)
    assert sw.nxt is None and sw.cur.dtype == torch.float16
    got = sw.test()
    assert got.dtype == torch.float16 and torch.equal(got.float(), ref)
    assert not sw._pending and not sw._noise_tiles


@pytest.mark.parametrize("world,hnm,state", [(2, HNM, "fp32x2"), (2, HNM, "fp16"), (2, 1, "fp32x2"), (3, 2, "fp16")])
def test_launcher_and_rank_entry_of_the_gpu_sweep(world, hnm, state, tmp_path):
    """The product launcher (launch.spawn_ranks: what `bench.py --gpus N` starts its ranks with) and the product rank
    entry (launch.run_sweep: what `bench.py --sweep` / tools/run_roi.py run per GPU), on CPU under gloo: every rank's
    rows equal the single-rank sweep bit for bit.  world > hnm (more ranks than tile rows: trailing ranks own nothing
    and must stay out of the strip exchange) used to hang the last rank that has rows."""
    import numpy as np
    from teramind_amd import launch
    torch.set_num_threads(4)
    ref = TileSweep(PathConfig(), StandInSampler(), None, gene_provider, hst=512, wst=768, hnm=hnm, wnm=WNM, total_epochs=T,
                    total_slc=SLC, device="cpu").test()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sweep_worker.py")
    rc = launch.spawn_ranks(world, [worker, str(tmp_path), str(hnm), str(WNM), state], timeout=600)
    assert rc == 0, f"rank processes failed / timed out (exit code {rc})"
    rows_seen = 0
    for r in range(world):
        r0, r1, w, backend, xbytes, nx = open(tmp_path / f"rank{r}.txt").read().split()
        r0, r1 = int(r0), int(r1)
        assert int(w) == world and backend == "gloo"
        got = torch.from_numpy(np.load(tmp_path / f"rank{r}.npy"))
        assert got.shape == (SLC * 2, (r1 - r0) * 256, WNM * 256)
        assert torch.equal(got, ref[:, r0 * 256:r1 * 256, :]), f"rank {r} rows [{r0},{r1}) differ from the single-rank sweep"
        if r1 > r0 and min(world, hnm) > 1:
            assert int(nx) > 0 and int(xbytes) > 0                   # it did exchange strips
        if r1 == r0:
            assert int(nx) == 0                                      # an empty rank stays out of the exchange
        rows_seen += r1 - r0
    assert rows_seen == hnm


def test_halo_dependence_is_real():
    """Sanity of the stand-in: without the exchange the two-rank result would differ."""
    sw = make_sweep(0, 1)
    a = sw._window(1, 0).clone()
    assert float((a[:32, 32:] != -1).float().mean()) == 1.0          # top halo of row 1 is row 0's state, not the -1 frame
    sw2 = TileSweep(PathConfig(), StandInSampler(), None, gene_provider, hst=512, wst=768, hnm=1, wnm=WNM,
                    total_epochs=T, total_slc=SLC)
    assert float((sw2._window(0, 0)[:32] == -1).float().mean()) == 1.0
    assert float((a[:, :32] == -1).float().mean()) == 1.0           # left frame: ROI border
