"""bench.py's process-group plumbing (not part of the product): one rank per GPU as the driver launches them, used for barriers and for
reducing the timing only.  The frame-sharded path has no data-path collective -- every frame is a key frame and every
slice resets its contexts (Source/CLI/Global.cpp:959-960, FFV1_Slice.cpp:180-197,274-275); the product shards inside one process, a lane per
device (rawcooked_amd/csrc/pipeline.hip, rcgpu_sequence_plan)."""
from __future__ import annotations


def max_over_ranks(dist, seconds: float, device) -> float:
    """Wall time of the slowest rank (the driver's contract for bench.py)."""
    if dist is None:
        return seconds
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def init(world: int, rank: int, local_rank: int, backend: str, device):
    """One process per GPU (bench.py, launched by torch.distributed.run): RCCL ("nccl") on the GPU box, gloo in the CPU tests.
    Returns torch.distributed, or None for a single process."""
    if world <= 1:
        return None
    import os
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    kw = {"device_id": device} if backend == "nccl" else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def barrier(dist) -> None:
    if dist is not None:
        dist.barrier()


def gather_floats(dist, values, device):
    """Every rank's list of floats on every rank: [[rank 0's], [rank 1's], ...] (the per-rank rows of a multi-GPU report).  All ranks call it."""
    vals = [float(v) for v in values]
    if dist is None:
        return [vals]
    import torch
    t = torch.tensor(vals, dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[float(x) for x in o.tolist()] for o in out]


def timed_steps(dist, device, step, steps: int, warmup: int, synchronize, mark=None) -> float:
    """The driver's timing contract: `warmup` untimed steps, then exactly `steps` steps bracketed by barrier + device synchronize on
    both sides; the result is the slowest rank's wall time.  `mark(k)`, if given, is called with -1 when the clock starts and with k
    after timed step k has been issued (bench.py reads the host clock there: the per-step times of the line's `step_ms`)."""
    import time
    for _ in range(max(0, warmup)):
        step()
    synchronize()
    barrier(dist)
    synchronize()
    t0 = time.perf_counter()
    if mark:
        mark(-1)
    for k in range(steps):
        step()
        if mark:
            mark(k)
    synchronize()
    barrier(dist)
    dt = time.perf_counter() - t0
    return max_over_ranks(dist, dt, device) if dist is not None else dt


def finish(dist) -> None:
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
