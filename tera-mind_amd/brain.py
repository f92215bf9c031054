"""Tiled whole-brain / ROI sweep with the diffusion state resident in device memory.

Replaces the data plane of the reference inference driver (test_brn.py:124-295 `Tester`,
utils/MBADataset_tst.py:91-123 `_pad_im`): there every diffusion step writes each 256x256x100
tile to a zarr .zip as fp16 and the next step re-reads the 3x3 neighbourhood of every tile to
assemble its 32-px halo; ranks are synchronised with three barriers per step around the
filesystem.  Here:

  * each rank owns a contiguous block of tile rows (tiles.row_block_partition) and keeps the
    state of its rows, plus a 32-px frame, as one canvas tensor [C, rows*256+64, wnm*256+64];
  * a step reads tile windows straight out of the canvas (the halo is already in place),
    runs the reference-shaped `sampler.sample(..., idx=T-epoch-1)` per tile batch and writes
    the centre of every tile into the second canvas, rounded through fp16 exactly where the
    reference casts (`out.half()`, test_brn.py:222);
  * after the step the top / bottom 32-px strips are exchanged with ranks r-1 / r+1
    (torch.distributed P2P: RCCL send/recv over xGMI on GPUs, gloo on CPU) -- this single
    exchange is both the halo "all-gather" and the step barrier;
  * outside the ROI the frame stays at -1 (MBADataset_tst.py:95).

The compute is injected (`sampler`, `model`) so that the distributed / indexing logic is
testable on CPU with gloo and a stand-in model; on a GPU box they are
teramind_amd.diffusion.SpacedDiffusionBeatGans and teramind_amd.unet.BeatGANsUNetModel.
"""
from typing import Callable, Optional

import torch

from . import tiles
from .config import PathConfig, Z_PAD

PAD = 32                     # halo = patch_size // 2 (test_brn.py:285 `pad=conf.patch_size // 2`)


class TileSweep:
    def __init__(self, conf: PathConfig, sampler, model, gene_provider: Callable[[int, int], torch.Tensor],
                 hst: int = 256, wst: int = 256, hnm: int = 32, wnm: int = 32, total_epochs: int = 15,
                 total_slc: int = 50, device="cpu", rank: int = 0, world: int = 1, batch_tiles: int = 1,
                 group=None, init: str = "reference", noise_provider: Optional[Callable] = None,
                 state: str = "fp32x2", share_halo: bool = False, batch_rows: int = 1, cache_level0: bool = False,
                 z_group: Optional[int] = None):
        """hst/wst/hnm/wnm/total_epochs mirror the test_brn CLI (test_brn.py:302-334).
        gene_provider(row, col) -> dense [20, 20, (total_slc + 2*zpad) * 500] gene tile (already
        block-summed and z-padded like MBADataset_tst._getgene/_pad_gn) for ABSOLUTE tile
        (row, col).  init: 'reference' = LCG-seeded CPU randn per tile; 'device' = torch.randn
        on the device seeded per tile (fast, not the reference's stream).
        state: 'fp32x2' = two fp32 canvases (read / write), 8 B per state element;
               'fp16'   = ONE float16 canvas, 2 B per element -- the layout that holds the whole brain
               (118 404 tiles x 256^2 x 100 = 1.55 TB) in the 8 x 288 GB of one node (SURVEY.md 8e capacity
               note).  Every value the reference keeps between steps is float16 (test_brn.py:222), so the
               canvas is lossless from step 1 on; step 0's float32 noise is never stored -- a three-row band of
               noise tiles is regenerated from the LCG seeds as the sweep moves down -- and new tile rows are
               committed one row late, once the row below has consumed the old halo.  Results are
               bit-identical to 'fp32x2'.
        share_halo: the tiles of one model call (`batch_tiles` consecutive tiles of a tile row) are handed over as ONE
               window of 256 * k + 64 columns, i.e. a (P + 1) x (k P + 1) encoder patch grid instead of k grids of
               (P + 1) x (P + 1): the reference runs the encoder twice on every patch column two neighbouring tiles
               share (the right halo of one is the left interior edge of the next, on the same 64-px grid), here it
               runs once -- 165 instead of 200 encoder patches per z-chunk at k = 8.  Encoder patches are computed in
               isolation (SURVEY 8e "global equivalence"), so the result is the same bit for bit PROVIDED the gene
               tiles agree where they overlap (they are cut with overlap from one gene map,
               utils/MBADataset_tst.py:65-91; `consistent_gene_provider` has that property, a per-tile seeded
               provider does not).  The window's genes: every tile's own 16 x 16 interior cells, the outer halo
               cells from the window's edge tiles.  batch_rows (share_halo only): tile rows per window -- the halo
               ROWS between them are shared the same way ((4 r + 1) x (4 k + 1) encoder patches for r x k tiles).
        z_group: images (z-chunks of tiles / windows) per MODEL call.  A call of the sweep holds n_z x (tiles or windows) images
               that never exchange data (test_brn.py:188-197,219-221); the model's workspace is proportional to the images
               of a call (twelve encoder skip tensors, the RNA pyramid and the widest decoder block's scratch live for the
               call: 8.8 GiB per tile, 123.8 GiB for a 4 x 4 window in the 16-bit modes).  z_group = g runs a call's images
               g at a time through the sampler -- the same kernels on the same per-image data, bit-identical -- so that
               large shared-halo windows fit beside the whole-brain canvas (a 4 x 4 window at g = 5: ~25 GiB).
        cache_level0: keep level 0 of the RNA conditioning (gene attention -> down_z -> Upsample, unet_ours.py:298-310: the
               part that reads the gene counts) of every model call of a step and reuse it in the following steps -- the
               genes of a tile are the same at each of the T steps (test_brn.py:232-255 re-reads and re-embeds them every
               step).  59 KB per encoder patch (37 MB per tile, 27 MB with 4 x 4 shared-halo windows): an ROI-scale option,
               not a whole-brain one.  Bit-identical (model.precompute_rna_level0 / tm_unet_forward_level0)."""
        if state not in ("fp32x2", "fp16"):
            raise ValueError(f"state {state!r}")
        self.state = state
        if conf.rna_slc not in (1, 4, 8, 16):
            raise NotImplementedError("rna_slc must be 1, 4, 8 or 16")
        if total_slc % max(1, conf.rna_slc // 2):
            raise ValueError(f"total_slc {total_slc} is not a multiple of the z chunk {conf.rna_slc // 2} "
                             f"(the reference uses tiles.state_slices(rna_slc) = {tiles.state_slices(conf.rna_slc)})")
        self.conf, self.sampler, self.model, self.gene = conf, sampler, model, gene_provider
        self.hnm, self.wnm, self.T = hnm, wnm, total_epochs
        self.row0, self.col0 = hst // tiles.TILE, wst // tiles.TILE          # MBADataset_tst.py:33
        self.total_slc, self.n_stain = total_slc, conf.n_stain
        self.chn = total_slc * conf.n_stain
        self.dev, self.rank, self.world, self.group = torch.device(device), rank, world, group
        self.batch_tiles, self.init, self.noise_provider = batch_tiles, init, noise_provider
        self.share_halo = bool(share_halo)
        self.batch_rows = max(1, int(batch_rows)) if self.share_halo else 1
        self.cache_level0 = bool(cache_level0) and hasattr(model, "precompute_rna_level0")
        self.z_group = None if not z_group else max(1, int(z_group))
        self._level0 = {}
        self._overlap_checked = set()                    # share_halo windows whose gene tiles were checked for agreement
        self.r0, self.r1 = tiles.row_block_partition(hnm, world)[rank]
        self.nrows = self.r1 - self.r0
        H = self.nrows * tiles.TILE + 2 * PAD
        W = wnm * tiles.TILE + 2 * PAD
        self.epoch = 0
        self._pending, self._noise_tiles, self._noise_cap = {}, {}, 18
        # halo-exchange accounting (bench.py --sweep): seconds (device-synchronised when time_exchange is set), bytes
        # sent + received by this rank, number of exchanges
        self.time_exchange, self.exchange_s, self.exchange_bytes, self.exchanges = False, 0.0, 0, 0
        if state == "fp16":
            self.cur = torch.full((self.chn, H, W), -1.0, dtype=torch.float16, device=self.dev)
            self.nxt = None
            return
        # two canvases (read: step e, write: step e+1); frame initialised to -1
        self.cur = torch.full((self.chn, H, W), -1.0, dtype=torch.float32, device=self.dev)
        self.nxt = torch.full((self.chn, H, W), -1.0, dtype=torch.float32, device=self.dev)
        self._fill_initial_noise()
        self._exchange(self.cur)

    # ---- state ------------------------------------------------------------------------------
    def _centre(self, canvas, lr: int, c: int):
        y, x = PAD + lr * tiles.TILE, PAD + c * tiles.TILE
        return canvas[:, y:y + tiles.TILE, x:x + tiles.TILE]

    def _noise_tile(self, row: int, col: int) -> torch.Tensor:
        """Step-0 tile 'h w c' at ABSOLUTE grid position (utils/MBADataset_tst.py:49-58)."""
        if self.init == "reference":
            return tiles.initial_noise_tile(row, col, self.chn)
        g = torch.Generator(device=self.dev)
        g.manual_seed(tiles.tile_noise_seed(row, col))
        return torch.randn((tiles.TILE, tiles.TILE, self.chn), generator=g, device=self.dev)

    def _fill_initial_noise(self):
        for lr in range(self.nrows):
            for c in range(self.wnm):
                t = self._noise_tile(self.row0 + self.r0 + lr, self.col0 + c)
                self._centre(self.cur, lr, c).copy_(t.permute(2, 0, 1))

    # ---- 'fp16' state: the window a call reads -------------------------------------------------------
    def _noise_tile_chw(self, row: int, col: int) -> torch.Tensor:
        """Step-0 tile [C, 256, 256] float32 at ROI position (row, col), through a small LRU (neighbouring windows share two
        tile columns; anything older is regenerated from its seed)."""
        key = (row, col)
        t = self._noise_tiles.pop(key, None)
        if t is None:
            t = self._noise_tile(self.row0 + row, self.col0 + col).permute(2, 0, 1).contiguous()
        self._noise_tiles[key] = t                                  # most recent last
        while len(self._noise_tiles) > self._noise_cap:
            self._noise_tiles.pop(next(iter(self._noise_tiles)))
        return t

    def _state_window(self, lr: int, nr: int, c0: int, k: int) -> torch.Tensor:
        """[C, 256 nr + 64, 256 k + 64] float32: local tile rows lr .. lr + nr - 1, columns c0 .. c0 + k - 1 with their 32-px
        halo, as the step reads them.  Later steps: a slice of the canvas -- the rows a call reads are not overwritten while
        their row group is in progress (new rows are committed one row late, _commit_rows).  Step 0: the LCG-seeded noise
        tiles of the window and of the ring of tiles around it (-1 outside the ROI, utils/MBADataset_tst.py:95-101),
        regenerated per window: nothing of the size of a tile row is held (a float32 row of the whole brain is 11 GB)."""
        T_ = tiles.TILE
        hh, ww = nr * T_ + 2 * PAD, k * T_ + 2 * PAD
        if self.epoch > 0:
            return self.cur[:, lr * T_:lr * T_ + hh, c0 * T_:c0 * T_ + ww].float()
        self._noise_cap = max(self._noise_cap, 2 * (nr + 2) * (k + 2))
        win = torch.full((self.chn, hh, ww), -1.0, dtype=torch.float32, device=self.dev)
        for i in range(-1, nr + 1):
            row = self.r0 + lr + i                                   # row inside the ROI (a neighbouring rank's row at the edges)
            if not 0 <= row < self.hnm:
                continue
            y0 = PAD + i * T_
            ya, yb = max(y0, 0), min(y0 + T_, hh)
            for j in range(-1, k + 1):
                col = c0 + j
                if not 0 <= col < self.wnm:
                    continue
                x0 = PAD + j * T_
                xa, xb = max(x0, 0), min(x0 + T_, ww)
                win[:, ya:yb, xa:xb] = self._noise_tile_chw(row, col)[:, ya - y0:yb - y0, xa - x0:xb - x0]
        return win

    def _commit_rows(self, upto: int):
        """Write the pending new rows <= upto into the canvas (their old values have been consumed)."""
        for lr in sorted(k for k in self._pending if k <= upto):
            self.cur[:, PAD + lr * tiles.TILE:PAD + (lr + 1) * tiles.TILE, PAD:-PAD].copy_(self._pending.pop(lr))

    def _neighbours(self):
        """(up, down): the nearest ranks above / below that own tile rows, or None.  With more ranks than tile rows
        (world > hnm: the trailing ranks of tiles.row_block_partition own nothing) the empty ranks take no part in the
        exchange, and the last rank with rows has no partner below it."""
        part = tiles.row_block_partition(self.hnm, self.world)
        has = [r1 > r0 for r0, r1 in part]
        up = next((r for r in range(self.rank - 1, -1, -1) if has[r]), None)
        down = next((r for r in range(self.rank + 1, self.world) if has[r]), None)
        return up, down

    def _exchange(self, canvas):
        """32-px strips to / from the neighbouring ranks (full canvas width: corners included).  RCCL send / recv over
        xGMI when the group's backend is nccl; with gloo the strips are staged through host memory (CPU tests, and
        multi-process runs that share one GPU)."""
        if self.world == 1 or self.nrows == 0:
            return
        import time
        import torch.distributed as dist
        up, down = self._neighbours()
        if up is None and down is None:
            return
        timed = self.time_exchange and canvas.is_cuda
        if timed:
            torch.cuda.synchronize(canvas.device)
        t0 = time.perf_counter()
        via_host = canvas.is_cuda and dist.get_backend(self.group) == "gloo"
        ops, bufs = [], []
        H = canvas.shape[1]
        for peer, src, dst in ((up, slice(PAD, 2 * PAD), slice(0, PAD)), (down, slice(H - 2 * PAD, H - PAD), slice(H - PAD, H))):
            if peer is None:
                continue
            send = canvas[:, src, :].contiguous()
            if via_host:
                send = send.cpu()
            recv = torch.empty_like(send)
            ops += [dist.P2POp(dist.isend, send, peer, self.group), dist.P2POp(dist.irecv, recv, peer, self.group)]
            bufs.append((recv, dst))
            self.exchange_bytes += 2 * send.numel() * send.element_size()
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        for recv, sl in bufs:
            canvas[:, sl, :].copy_(recv)
        if timed:
            torch.cuda.synchronize(canvas.device)
        self.exchange_s += time.perf_counter() - t0
        self.exchanges += 1

    # ---- one diffusion step over the rank's tiles ----------------------------------------------
    def _window(self, lr: int, c: int) -> torch.Tensor:
        y, x = lr * tiles.TILE, c * tiles.TILE
        if self.state == "fp16":
            return self._state_window(lr, 1, c, 1).permute(1, 2, 0)
        return self.cur[:, y:y + tiles.TILE + 2 * PAD, x:x + tiles.TILE + 2 * PAD].permute(1, 2, 0)     # 'h w c'

    def _run_row_window(self, lr: int, c0: int, k: int, epoch: int, nr: int = 1):
        """share_halo: tiles (lr .. lr + nr - 1) x (c0 .. c0 + k - 1) as one window -> [C, 256 nr, 256 k] float32 result."""
        conf = self.conf
        x0, wpx = c0 * tiles.TILE, k * tiles.TILE + 2 * PAD
        if self.state == "fp16":
            win = self._state_window(lr, nr, c0, k)
        else:
            y = lr * tiles.TILE
            win = self.cur[:, y:y + nr * tiles.TILE + 2 * PAD, x0:x0 + wpx]
        ps = conf.patch_size
        xz = tiles.zchunk_state(win.permute(1, 2, 0)[None], self.total_slc, conf.rna_slc)
        shape = (xz.shape[0], xz.shape[3], xz.shape[1] - ps, xz.shape[2] - ps)
        x = tiles.patchify_hwc(xz, ps, True)
        key = ("w", lr, c0, nr, k)
        rna = self._level0.get(key)
        if rna is None:
            blk = ps // conf.gn_sz                                            # pixels per gene cell
            hc, ic = PAD // blk, tiles.TILE // blk                           # halo / interior cells per tile side
            bands = []
            grid = [[self.gene(self.row0 + self.r0 + lr + i, self.col0 + c0 + j).to(self.dev) for j in range(k)] for i in range(nr)]
            if key not in self._overlap_checked:
                # One window per call keeps every tile's interior cells and the edge tiles' halo: that equals the reference's
                # per-tile calls (utils/MBADataset_tst.py:65-89: every tile brings its own 4-cell halo) only if neighbouring
                # tiles AGREE on the 2 * hc cells they both hold -- true for tiles cut with overlap from one gene map, checked
                # here once per window instead of assumed.
                self._check_gene_overlap(grid, lr, c0, hc, ic)
                self._overlap_checked.add(key)
            for i in range(nr):                                               # one band of cell rows per tile row
                gts = grid[i]
                band = torch.cat([gts[0][:, :hc]] + [g[:, hc:hc + ic] for g in gts] + [gts[-1][:, hc + ic:]], dim=1)
                r0 = 0 if i == 0 else hc                                      # outer halo rows from the window's edge tiles
                r1 = hc + ic + (hc if i == nr - 1 else 0)
                bands.append(band[r0:r1])
            rna = tiles.patchify_hwc(tiles.zchunk_rna(torch.cat(bands, dim=0)[None], conf.rna_slc), conf.gn_sz, False)
            rna = self._remember_level0(key, rna, shape)
        out = self._sample(shape, x, rna, ps, epoch)
        return tiles.regroup_output(out, 1, self.n_stain)[0]

    def _sample(self, shape, x, rna, ps, epoch):
        """sampler.sample (test_brn.py:209-217) on the images of a call, z_group at a time: images are independent, so the
        groups' results concatenated equal the one call's bit for bit; the model workspace is that of one group."""
        n = shape[0]
        g = self.z_group
        if not g or g >= n:
            return self.sampler.sample(model=self.model, shape=shape, imgs=x, noise=x, r_start=rna, patch_size=ps,
                                       idx=self.T - epoch - 1, model_kwargs=None)
        pp = x.shape[0] // n                                              # padded patches per image
        outs = []
        for i0 in range(0, n, g):
            i1 = min(n, i0 + g)
            if hasattr(rna, "buf"):                                           # RnaLevel0: patch-major buffer of equal slices
                per = rna.buf.numel() // (rna.b * rna.p1 * rna.p2)
                r = type(rna)(rna.buf[i0 * pp * per:i1 * pp * per], i1 - i0, rna.p1, rna.p2)
            else:
                r = rna[i0 * pp:i1 * pp]
            outs.append(self.sampler.sample(model=self.model, shape=(i1 - i0,) + tuple(shape[1:]), imgs=x[i0 * pp:i1 * pp],
                                            noise=x[i0 * pp:i1 * pp], r_start=r, patch_size=ps, idx=self.T - epoch - 1,
                                            model_kwargs=None))
        return torch.cat(outs)

    def _check_gene_overlap(self, grid, lr, c0, hc, ic):
        """share_halo precondition: the gene tiles of a window agree on the cells neighbouring tiles both hold."""
        nr, k = len(grid), len(grid[0])
        name = lambda i, j: f"tile (row {self.row0 + self.r0 + lr + i}, col {self.col0 + c0 + j})"
        for i in range(nr):
            for j in range(k):
                g = grid[i][j]
                if j + 1 < k and not torch.equal(g[:, ic:ic + 2 * hc], grid[i][j + 1][:, :2 * hc]):
                    raise ValueError(f"share_halo: gene {name(i, j)} and {name(i, j + 1)} disagree on the {2 * hc} cell columns they "
                                     "share; one window per call would drop one tile's version and differ from the reference's "
                                     "per-tile result -- use share_halo=False for tiles that were not cut from one gene map")
                if i + 1 < nr and not torch.equal(g[ic:ic + 2 * hc, :], grid[i + 1][j][:2 * hc, :]):
                    raise ValueError(f"share_halo: gene {name(i, j)} and {name(i + 1, j)} disagree on the {2 * hc} cell rows they "
                                     "share; one window per call would drop one tile's version and differ from the reference's "
                                     "per-tile result -- use share_halo=False for tiles that were not cut from one gene map")

    def _remember_level0(self, key, rna, shape):
        """cache_level0: level 0 of the RNA conditioning of this call's genes, kept for the following steps."""
        if not self.cache_level0:
            return rna
        l0 = self.model.precompute_rna_level0(rna, shape[0], imgs=torch.empty(tuple(shape), device="meta"),
                                              patch_size=self.conf.patch_size)
        self._level0[key] = l0
        return l0

    def run_batch(self, tile_list, epoch: int):
        """Tester._run_batch (test_brn.py:174-226) for a list of (local_row, col) tiles."""
        conf = self.conf
        if self.share_halo and len(tile_list) > 1:
            lr, c0 = tile_list[0]
            nr = tile_list[-1][0] - lr + 1
            k = len(tile_list) // nr
            if [(lr + i, c0 + j) for i in range(nr) for j in range(k)] != list(tile_list):
                raise ValueError("share_halo: the tiles of a call must form a rectangle of consecutive rows and columns")
            out = self._run_row_window(lr, c0, k, epoch, nr)
            x0, x1 = c0 * tiles.TILE, (c0 + k) * tiles.TILE
            for i in range(nr):
                o = out[:, i * tiles.TILE:(i + 1) * tiles.TILE]
                if self.state == "fp16":
                    if lr + i not in self._pending:
                        self._pending[lr + i] = torch.empty((self.chn, tiles.TILE, self.wnm * tiles.TILE), dtype=torch.float16,
                                                            device=self.dev)
                    self._pending[lr + i][:, :, x0:x1].copy_(o.half())        # test_brn.py:222
                else:
                    y = PAD + (lr + i) * tiles.TILE
                    self.nxt[:, y:y + tiles.TILE, PAD + x0:PAD + x1].copy_(o.half().float())
            return
        tile_hwc = torch.stack([self._window(lr, c) for lr, c in tile_list])
        key = ("t",) + tuple(tile_list)
        rna = self._level0.get(key)
        if rna is None:
            rna_hwc = torch.stack([self.gene(self.row0 + self.r0 + lr, self.col0 + c).to(self.dev) for lr, c in tile_list])
            x, rna, shape = tiles.run_batch_inputs(tile_hwc, rna_hwc, conf.patch_size, conf.gn_sz, self.total_slc, conf.rna_slc)
            rna = self._remember_level0(key, rna, shape)
        else:
            xz = tiles.zchunk_state(tile_hwc, self.total_slc, conf.rna_slc)
            shape = (xz.shape[0], xz.shape[3], xz.shape[1] - conf.patch_size, xz.shape[2] - conf.patch_size)
            x = tiles.patchify_hwc(xz, conf.patch_size, True)
        out = self._sample(shape, x, rna, conf.patch_size, epoch)
        out = tiles.regroup_output(out, len(tile_list), self.n_stain)
        if self.state == "fp16":
            out = out.half()                                                  # test_brn.py:222
            for k, (lr, c) in enumerate(tile_list):
                if lr not in self._pending:
                    self._pending[lr] = torch.empty((self.chn, tiles.TILE, self.wnm * tiles.TILE), dtype=torch.float16,
                                                    device=self.dev)
                self._pending[lr][:, :, c * tiles.TILE:(c + 1) * tiles.TILE].copy_(out[k])
            return
        out = out.half().float()                                              # test_brn.py:222
        for k, (lr, c) in enumerate(tile_list):
            self._centre(self.nxt, lr, c).copy_(out[k])

    def step(self):
        todo = [(lr, c) for lr in range(self.nrows) for c in range(self.wnm)]
        if self.share_halo:                              # rectangles of batch_rows x batch_tiles tiles (ragged at the edges)
            batches = [[(lr, c) for lr in range(l0, min(l0 + self.batch_rows, self.nrows))
                        for c in range(c0, min(c0 + self.batch_tiles, self.wnm))]
                       for l0 in range(0, self.nrows, self.batch_rows) for c0 in range(0, self.wnm, self.batch_tiles)]
        else:
            batches = [todo[i:i + self.batch_tiles] for i in range(0, len(todo), self.batch_tiles)]
        for batch in batches:
            self.run_batch(batch, self.epoch)
            if self.state == "fp16":
                # a row group is complete once its call reaches the last column; the old values of its LAST row are still
                # read by the row group below (top halo), the old values of the row above the group by its remaining calls
                self._commit_rows(batch[-1][0] - 1 if batch[-1][1] == self.wnm - 1 else batch[0][0] - 2)
        if self.state == "fp16":
            self._commit_rows(self.nrows)
            self._noise_tiles = {}
            self._exchange(self.cur)
            self.epoch += 1
            return
        self._exchange(self.nxt)
        self.cur, self.nxt = self.nxt, self.cur
        self.epoch += 1

    def test(self):
        """Tester.test (test_brn.py:232-273): one diffusion step per epoch."""
        while self.epoch < self.T:
            self.step()
        return self.local_state()

    # ---- results -------------------------------------------------------------------------------
    def local_state(self) -> torch.Tensor:
        """[C, nrows*256, wnm*256] state of this rank's rows (without the frame)."""
        return self.cur[:, PAD:self.cur.shape[1] - PAD, PAD:self.cur.shape[2] - PAD]

    def tile(self, lr: int, c: int) -> torch.Tensor:
        """[C, 256, 256] -- what the reference stores as '{name}.zip' (fp16) after each step."""
        return self._centre(self.cur, lr, c)

    # ---- interoperability with the reference's per-step tile directories -------------------------
    def step_dir(self, out_dir, epoch: Optional[int] = None) -> str:
        """'{out_dir}_{epoch}': where the reference keeps the state after `epoch` steps (test_brn.py:247)."""
        return f"{out_dir}_{self.epoch if epoch is None else epoch}"

    def save_step(self, out_dir, compressor: Optional[str] = None) -> str:
        """Write this rank's tiles as the reference would have after `self.epoch` steps: one zarr-v2 .zip per
        tile, float16 [(stain z), 256, 256], named '{r0}_{r1}_{c0}_{c1}.zip' (test_brn.py:222-226)."""
        import os
        from . import formats
        if self.epoch == 0:
            raise ValueError("step 0 is noise regenerated from the tile seeds; the reference stores no '_0' directory")
        d = self.step_dir(out_dir)
        os.makedirs(d, exist_ok=True)
        for lr in range(self.nrows):
            band = self.cur[:, PAD + lr * tiles.TILE:PAD + (lr + 1) * tiles.TILE, PAD:-PAD].half().cpu().numpy()
            for c in range(self.wnm):
                name = tiles.state_tile_name(self.row0 + self.r0 + lr, self.col0 + c)
                formats.write_state_tile(os.path.join(d, name + ".zip"), band[:, :, c * tiles.TILE:(c + 1) * tiles.TILE],
                                         compressor)
        return d

    def load_step(self, out_dir, epoch: int):
        """Resume from a reference-format step directory (`--cur_epoch`, test_brn.py:290-292): read this rank's
        tiles of '{out_dir}_{epoch}', then exchange halos.  Works on directories written by the reference
        (Blosc-lz4 chunks) and by save_step."""
        import os
        import numpy as np
        from . import formats
        if not 0 < epoch <= self.T:
            raise ValueError(f"epoch {epoch} outside 1..{self.T}")
        d = self.step_dir(out_dir, epoch)
        self.cur.fill_(-1.0)
        self._pending, self._noise_tiles = {}, {}
        for lr in range(self.nrows):
            for c in range(self.wnm):
                name = tiles.state_tile_name(self.row0 + self.r0 + lr, self.col0 + c)
                a = formats.read_state_tile(os.path.join(d, name + ".zip"))
                if a.shape != (self.chn, tiles.TILE, tiles.TILE):
                    raise ValueError(f"{name}.zip: shape {a.shape}, expected {(self.chn, tiles.TILE, tiles.TILE)}")
                self._centre(self.cur, lr, c).copy_(torch.from_numpy(a.astype(np.float32)))
        self.epoch = epoch
        self._exchange(self.cur)


def synthetic_gene_provider(conf: PathConfig, total_slc: int = 50, density: float = 0.02, device="cpu"):
    """Config-3 synthetic genes: a [20, 20, 26000] tile seeded by the tile
    index; zero in the z-padding slices like MBADataset_tst._getgene.  total_slc = GENE slices (50 in the
    reference data, also for the 8- / 16-slice models whose state has 48)."""
    from . import synth
    zpad = Z_PAD[conf.rna_slc]
    cells = (tiles.TILE + 2 * PAD) // (conf.patch_size // conf.gn_sz)

    def provider(row: int, col: int) -> torch.Tensor:
        core = synth.gene_counts(f"gene/{row}/{col}", (cells, cells, total_slc * tiles.GENES), 0, density)
        if zpad:
            z = torch.zeros((cells, cells, zpad * tiles.GENES))
            core = torch.cat((z, core, z), dim=-1)
        return core.to(device)
    return provider


def device_gene_provider(cfg: PathConfig, dev, total_slc: int = 50, density: float = 0.02):
    """Synthetic gene tiles generated and kept on the device: [20, 20, (total_slc + 2 zpad) * 500] sparse non-negative
    integer counts, a pure function of the absolute tile position (so every rank sees the same tile for (row, col))."""
    cache = {}
    zpad = Z_PAD[cfg.rna_slc] * tiles.GENES
    cells = (tiles.TILE + 2 * PAD) // (cfg.patch_size // cfg.gn_sz)

    def provider(row, col):
        if (row, col) not in cache:
            g = torch.Generator(device=dev)
            g.manual_seed(1_000_003 * row + col)
            u = torch.rand((cells, cells, total_slc * tiles.GENES), generator=g, device=dev)
            core = torch.where(u < density, torch.floor(u * (3.0 / density)) + 1.0, torch.zeros_like(u))
            if zpad:
                z = torch.zeros((cells, cells, zpad), device=dev)
                core = torch.cat((z, core, z), dim=-1)
            cache[(row, col)] = core
        return cache[(row, col)]
    return provider


def consistent_gene_provider(cfg: PathConfig, dev, total_slc: int = 50, density: float = 0.02, max_blocks: int = 0,
                             max_tiles: int = 0):
    """Synthetic gene tiles that AGREE where neighbouring tiles overlap, like tiles cut with overlap from one gene map
    (utils/MBADataset_tst.py:65-91): the map is a pure function of the absolute 16 x 16-cell block (seeded per block), a
    tile's [20, 20, G] grid is assembled from its own block and the 2-cell rims of its eight neighbours.  Kept on the
    device; the precondition of TileSweep(share_halo=True).  max_blocks / max_tiles (> 0): bound the two caches to that many
    entries (oldest evicted first; a block is 25.6 MB, a dense tile 41.6 MB) -- for sweeps as wide as the whole brain, where
    the default five-row band of blocks alone would take 53 GB."""
    blocks, cache = {}, {}
    zpad = Z_PAD[cfg.rna_slc] * tiles.GENES
    blk = cfg.patch_size // cfg.gn_sz
    ic, hc = tiles.TILE // blk, PAD // blk

    def block(R, C):
        if (R, C) not in blocks:
            g = torch.Generator(device=dev)
            g.manual_seed(2_000_003 * (R + 7) + (C + 7))
            u = torch.rand((ic, ic, total_slc * tiles.GENES), generator=g, device=dev)
            blocks[(R, C)] = torch.where(u < density, torch.floor(u * (3.0 / density)) + 1.0, torch.zeros_like(u))
            for k in [k for k in blocks if abs(k[0] - R) > 2]:      # the sweep moves along the rows: keep a five-row band
                del blocks[k]
            while max_blocks > 0 and len(blocks) > max_blocks:
                del blocks[next(iter(blocks))]                      # insertion order: the oldest first
        return blocks[(R, C)]

    def provider(row, col):
        if (row, col) not in cache:
            rs = [(row - 1, slice(ic - hc, ic)), (row, slice(0, ic)), (row + 1, slice(0, hc))]
            cs = [(col - 1, slice(ic - hc, ic)), (col, slice(0, ic)), (col + 1, slice(0, hc))]
            core = torch.cat([torch.cat([block(R, C)[sr, sc] for C, sc in cs], dim=1) for R, sr in rs], dim=0)
            if zpad:
                z = torch.zeros((ic + 2 * hc, ic + 2 * hc, zpad), device=dev)
                core = torch.cat((z, core, z), dim=-1)
            for k in [k for k in cache if k[0] != row]:             # one tile row of dense grids (41.6 MB each) at a time
                del cache[k]
            cache[(row, col)] = core
            while max_tiles > 0 and len(cache) > max_tiles:
                del cache[next(iter(cache))]
        return cache[(row, col)]
    return provider


class GeneTileDir:
    """gene_provider over a directory of the reference's gene tiles (test_brn.py:51-70 `gn_sublst` names,
    utils/MBADataset_tst.py:65-91,140-154): '{r0}_{r1}_{c0}_{c1}_{R0}_{R1}_{C0}_{C1}.npz' COO archives of the
    tile padded by 128 px.  The COO arrays are uploaded once and stay resident (a few MB per tile instead of
    the 41.6 MB dense grid); every call re-densifies on the device with `tm_gene_tile_dense`
    (block sum + halo shift + crop + z padding in one scatter pass)."""

    def __init__(self, gdir, conf: PathConfig, device, total_slc: int = 50, keep_resident: bool = True):
        """total_slc: slices in the gene files (50 for every config; the 8- / 16-slice models keep 48 STATE slices,
        tiles.state_slices, but read all 50 gene slices plus z padding)."""
        import os
        self.gdir, self.conf, self.dev = str(gdir), conf, torch.device(device)
        self.gblk = conf.patch_size // conf.gn_sz                      # test_brn.py:282 `_blk`
        self.gsz = (tiles.TILE + 2 * PAD) // self.gblk
        self.chan_in = total_slc * tiles.GENES
        self.zpad_ch = Z_PAD[conf.rna_slc] * tiles.GENES
        self.keep, self.cache = keep_resident, {}
        self._os = os

    def path(self, row: int, col: int) -> str:
        r0, c0, half = row * tiles.TILE, col * tiles.TILE, tiles.TILE // 2
        v = (r0, r0 + tiles.TILE, c0, c0 + tiles.TILE, r0 - half, r0 + tiles.TILE + half, c0 - half, c0 + tiles.TILE + half)
        return self._os.path.join(self.gdir, "_".join(str(x) for x in v) + ".npz")

    def _load(self, row: int, col: int):
        import numpy as np
        from . import formats
        p = self.path(row, col)
        data, coords, shape = formats.read_gene_npz(p)
        if len(shape) != 3 or shape[2] != self.chan_in:
            raise ValueError(f"{p}: gene tile shape {shape}, expected [*, *, {self.chan_in}]")
        roi, roio = formats.parse_gene_tile_name(p)
        if shape[0] != roio[1] - roio[0] or shape[1] != roio[3] - roio[2]:
            raise ValueError(f"{p}: array extent {shape[:2]} does not match the padded ROI in its name")
        sh, sw = formats.gene_tile_shift(roi, roio, self.gblk, PAD)
        crd = torch.from_numpy(np.ascontiguousarray(coords.astype(np.int32))).to(self.dev)
        dat = torch.from_numpy(np.ascontiguousarray(np.asarray(data).astype(float).astype(np.float32))).to(self.dev)
        return crd, dat, sh, sw

    def __call__(self, row: int, col: int) -> torch.Tensor:
        from . import _lib
        ent = self.cache.get((row, col))
        if ent is None:
            ent = self._load(row, col)
            if self.keep:
                self.cache[(row, col)] = ent
        crd, dat, sh, sw = ent
        out = torch.empty((self.gsz, self.gsz, self.chan_in + 2 * self.zpad_ch), dtype=torch.float32, device=self.dev)
        _lib.check(_lib.lib().tm_gene_tile_dense(_lib.ptr(crd), _lib.ptr(dat), dat.numel(), self.gblk, sh, sw, self.gsz,
                                                 self.chan_in, self.zpad_ch, _lib.ptr(out), _lib.current_stream_ptr()),
                   "tm_gene_tile_dense")
        return out
