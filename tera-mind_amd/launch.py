"""One process per GPU: the launcher and the per-rank entry of the row-sharded tile sweep.

Replaces the process scaffold of the reference inference driver -- `mp.spawn(main, nprocs=world_size)`
(test_brn.py:349-351), `ddp_setup` (:26-35: MASTER_ADDR / MASTER_PORT, `init_process_group("nccl")`), the DDP wrap whose
only effect at inference is the construction-time parameter broadcast (:149) and `main(rank, world_size, T, conf, args)`
(:276-295).  Here:

  * `spawn_ranks` starts N FRESH interpreter processes (no fork of a parent that may have touched the GPU, no exec from
    one), each with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in its environment -- the
    same contract `python -m torch.distributed.run` provides, so a worker cannot tell which of the two launched it;
  * `init_distributed` is the rank-side counterpart of `ddp_setup`;
  * `run_sweep` is `main` + `Tester.test`: rank r owns a contiguous block of tile rows (brain.TileSweep), the packed
    weight arena is broadcast from rank 0 once, every diffusion step ends with the 32-px strip exchange with the
    neighbouring ranks (RCCL send / recv over xGMI), and the step times / exchange times / bytes are returned.

The compute objects (sampler, model, gene provider) are injected, so the same function runs on CPU under gloo with a
stand-in model (tests/test_dist_gloo.py) and on GPUs under nccl with the HIP model (bench.py --sweep, tools/run_roi.py).
"""
import os
import socket
import subprocess
import sys
import time
from typing import Callable, List, Optional, Sequence


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def dist_env():
    """(rank, local_rank, world) from the launcher's environment; (0, 0, 1) when not launched as a rank."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))


def launched_as_rank() -> bool:
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def spawn_ranks(n: int, argv: Sequence[str], port: Optional[int] = None, env: Optional[dict] = None,
                timeout: Optional[float] = None, python: Optional[str] = None) -> int:
    """Start `n` worker processes `python argv...`, rank r with RANK=LOCAL_RANK=r, and wait for them.
    Returns 0 when every rank exited 0, otherwise the first non-zero exit code (the remaining ranks of a failed job are
    terminated by PID).  stdout / stderr are inherited: rank 0 prints the job's result.
    Must be called before the calling process initialises a GPU (it never does: this module does not import torch)."""
    if n < 1:
        raise ValueError("n must be >= 1")
    port = port or free_port()
    procs: List[subprocess.Popen] = []
    for r in range(n):
        e = dict(os.environ)
        e.update(env or {})
        e.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                  "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        # The caller's environment decides; only when it says nothing, pick dmabuf IPC: this image's environment notes state that
        # the host driver supports dmabuf IPC only and that RCCL / cross-process device-memory sharing fails with
        # `hipIpcGetMemHandle: invalid argument` under the legacy mode (the image itself exports the variable as 0).  Not
        # verified by a multi-GPU run of this repository: no multi-GPU box was available to it.
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([python or sys.executable] + list(argv), env=e))
    deadline = None if timeout is None else time.monotonic() + timeout
    rc = 0
    try:
        live = list(procs)
        while live:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code
            if rc != 0 or (deadline is not None and time.monotonic() > deadline):
                if rc == 0:
                    rc = 124                                     # timeout
                break
            time.sleep(0.05)
    finally:
        for p in procs:                                          # exact PIDs of the children started above
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return rc


def init_distributed(backend: str, device=None):
    """`ddp_setup` (test_brn.py:26-35) for a process started by spawn_ranks or torch.distributed.run.
    Returns (rank, local_rank, world); no process group is created for world == 1."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = dist_env()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = torch.device(device)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def broadcast_arena(model) -> int:
    """Rank 0's packed weight arena to every rank: ONE collective in place of DDP's construction-time parameter broadcast
    (test_brn.py:149).  RCCL on the device; with gloo (CPU tests, rehearsals that share one GPU) staged through host memory.
    Returns the number of bytes broadcast (0 when there is no process group)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1 or not hasattr(model, "arena"):
        return 0
    a = model.arena()
    if dist.get_backend() == "gloo" and a.is_cuda:
        h = a.cpu()
        dist.broadcast(h, src=0)
        a.copy_(h)
    else:
        dist.broadcast(a, src=0)
    return int(a.numel())


def reduce_max(values, device):
    """MAX over ranks of a few host floats (timings); returns them unchanged without a process group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return list(values)
    t = torch.tensor(list(values), dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t]


def run_sweep(conf, sampler, model, gene_provider: Callable, *, hnm: int, wnm: int, total_epochs: int, steps: int,
              warmup: int = 0, device="cpu", total_slc: int = 50, hst: int = 256, wst: int = 256, batch_tiles: int = 1,
              init: str = "device", state: str = "fp16", broadcast_weights: bool = True, time_exchange: bool = True,
              prefetch_genes: bool = True, on_step: Optional[Callable] = None,
              after_warmup: Optional[Callable] = None, holder: Optional[dict] = None, share_halo: bool = False,
              batch_rows: int = 1, cache_level0: bool = False, z_group: Optional[int] = None) -> dict:
    """`main` + `Tester.test` of the reference (test_brn.py:232-295) for this rank: build the rank's TileSweep, run
    `warmup` untimed then `steps` timed diffusion steps (each ending with the halo-strip exchange), and return
    {'sweep': TileSweep, 'dt': seconds of the timed steps on this rank (max over ranks when world > 1), 'step_s': [...],
     'warmup_s': [seconds of each warm-up step on this rank],
     'exchange_ms_per_step', 'exchange_bytes_per_step', 'world', 'backend', 'rows': (r0, r1)}.
    The timed region is bracketed by a barrier + device synchronise on both sides."""
    import torch
    import torch.distributed as dist
    from . import brain
    rank, _, world = dist_env()
    dist_on = world > 1 and dist.is_initialized()
    if not dist_on:
        rank, world = 0, 1
    dev = torch.device(device)
    cuda = dev.type == "cuda"
    if dist_on and broadcast_weights and model is not None:
        broadcast_arena(model)
    sw = brain.TileSweep(conf, sampler, model, gene_provider, hst=hst, wst=wst, hnm=hnm, wnm=wnm, total_epochs=total_epochs,
                         total_slc=total_slc, device=dev, rank=rank, world=world, batch_tiles=batch_tiles, init=init, state=state,
                         share_halo=share_halo, batch_rows=batch_rows, cache_level0=cache_level0, z_group=z_group)
    sw.time_exchange = time_exchange
    if holder is not None:
        holder["sweep"] = sw
    if prefetch_genes:                                           # gene tiles resident before the timed region
        for lr in range(sw.nrows):
            for c in range(wnm):
                gene_provider(sw.row0 + sw.r0 + lr, sw.col0 + c)

    def fence():
        if cuda:
            torch.cuda.synchronize(dev)
        if dist_on:
            dist.barrier()
        if cuda:
            torch.cuda.synchronize(dev)

    warm_s = []                                                  # the first step of a sweep also computes what later steps reuse
    for _ in range(warmup):
        t1 = time.perf_counter()
        sw.step()
        if cuda:
            torch.cuda.synchronize(dev)
        warm_s.append(time.perf_counter() - t1)
    fence()
    if after_warmup is not None:
        after_warmup()
    sw.exchange_s, sw.exchange_bytes, sw.exchanges = 0.0, 0, 0
    step_s = []
    if holder is not None:
        holder["step_s"] = step_s
    t0 = time.perf_counter()
    for k in range(steps):
        t1 = time.perf_counter()
        sw.step()
        if on_step is not None:
            if cuda:
                torch.cuda.synchronize(dev)
            step_s.append(time.perf_counter() - t1)
            on_step(sw, step_s[-1])
    fence()
    dt = time.perf_counter() - t0
    exch_ms = 1e3 * sw.exchange_s / max(1, steps)
    dt, exch_ms = reduce_max([dt, exch_ms], dev)
    return {"sweep": sw, "dt": dt, "step_s": step_s, "warmup_s": warm_s, "exchange_ms_per_step": exch_ms,
            "exchange_bytes_per_step": sw.exchange_bytes // max(1, steps), "world": dist.get_world_size() if dist_on else 1,
            "backend": dist.get_backend() if dist_on else "none", "rows": (sw.r0, sw.r1)}
