"""-m gpu: the multi-process GPU path on ONE GPU.  `bench.py --gpus 2 --rehearse` starts two rank processes with the
product launcher (teramind_amd.launch.spawn_ranks, no torchrun), both on cuda:0 with a gloo group: everything of the
N-GPU sweep runs -- per-rank models, weight-arena broadcast, row-block partition of the tile grid, the per-step 32-px
strip exchange, the timing reductions -- except RCCL itself (one GPU cannot host two RCCL ranks).  The ROI state after the
sweep must be bit-identical to the single-process run: compared through an exact, partition-independent integer digest."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--sweep", "--sweep-hnm", "2", "--sweep-wnm", "2", "--steps", "2", "--warmup", "0",
           "--no-cpu-baseline"] + extra
    env = dict(os.environ)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def test_two_rank_sweep_on_one_gpu_equals_single_process():
    one = _bench(["--gpus", "1"])
    two = _bench(["--gpus", "2", "--rehearse"])
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "strong"
    s1, s2 = one["sweep"], two["sweep"]
    assert s2["world_size_rccl"] == 2 and s2["backend"] == "gloo" and s2["rows_rank0"] == [0, 1]
    assert s2["exchange_bytes_per_step_per_rank"] == 2 * 100 * 32 * (2 * 256 + 64) * 2        # one fp16 strip sent + one received
    assert s2["weights_broadcast_bytes"] > 4e8
    assert s1["state_digest"] == s2["state_digest"] and s1["state_digest"][2] == 100 * 512 * 512
    assert two["roofline"]["frac"] <= 1.0 and two["roofline"]["launches_timed"] > 0


def test_default_bench_line_with_two_ranks_rehearsed():
    """The driver's form `bench.py --gpus N --steps K --warmup W` at N = 2 (self-spawned ranks): the weak-scaling
    configs[1] line plus the appended row-sharded sweep object."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse", "--steps", "2", "--warmup", "1", "--sweep-hnm", "2",
           "--sweep-wnm", "1", "--sweep-steps", "1"]
    env = dict(os.environ)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0)"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["dtype"] == "f32" and d["value"] > 0
    assert d["config"]["per_gpu_patches_per_step"] == 32 and 0 < d["roofline"]["frac"] <= 1.0
    assert d["sweep"]["world_size_rccl"] == 2 and d["sweep"]["scaling"] == "strong" and "cpu_baseline" not in d


def test_two_rank_sweep_over_rccl_equals_single_process():
    """The real thing, for the first lease that has two GPUs: `bench.py --gpus 2 --sweep` with one rank per GPU over RCCL
    (backend "nccl": arena broadcast on device, batch_isend_irecv strip exchange, on-device reductions).  Skipped on the
    one-GPU boxes this repository has been developed on -- where it has therefore never run."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    one = _bench(["--gpus", "1"])
    two = _bench(["--gpus", "2"])
    s1, s2 = one["sweep"], two["sweep"]
    assert s2["world_size_rccl"] == 2 and s2["backend"] == "nccl" and s2["rows_rank0"] == [0, 1]
    assert s1["state_digest"] == s2["state_digest"]
    assert s2["exchange_bytes_per_step_per_rank"] == 2 * 100 * 32 * (2 * 256 + 64) * 2
