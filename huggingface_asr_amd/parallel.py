"""Multi-GPU plumbing of the hot path: one process per GPU, `torch.distributed` (backend nccl = RCCL on ROCm,
gloo on CPU for tests).  The forward path shards by utterance and has NO exchange step, so the only collectives are the
barrier / MAX-reduction that bracket a timed region and the mean of per-rank losses for logging (the reference gathers its
logged losses the same way: src/utilities/training_utils.py:357-361 `_nested_gather(...).mean()`)."""
from __future__ import annotations

import os
import time

import torch
import torch.distributed as dist


COLLECTIVE_TIMEOUT_S = 120          # a rank that never arrives fails the others after two minutes (torch's default is ten: longer than the driver waits for a bench)


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend: str | None = None, device: torch.device | None = None):
    """Initialise the default process group from the torchrun environment (no-op for a single process)."""
    world, rank, local = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        import datetime
        dist.init_process_group(backend, timeout=datetime.timedelta(seconds=COLLECTIVE_TIMEOUT_S), **kw)
    return world, rank, local


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous, balanced [lo, hi) slice of `n_items` utterances for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device="cpu") -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def mean_over_ranks(x: torch.Tensor) -> torch.Tensor:
    """Mean of a per-rank scalar (e.g. the batch-mean CTC loss of every rank's shard)."""
    if not dist.is_initialized():
        return x
    y = x.detach().clone().float()
    dist.all_reduce(y, op=dist.ReduceOp.SUM)
    return y / dist.get_world_size()


def timed(step_fn, steps: int, sync=None, device="cpu") -> float:
    """Time exactly `steps` calls of step_fn between barriers (+ device sync on both sides); MAX over ranks (bench contract)."""
    sync = sync or (lambda: None)
    sync(); barrier(); sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    sync(); barrier(); sync()
    return max_over_ranks(time.perf_counter() - t0, device)
