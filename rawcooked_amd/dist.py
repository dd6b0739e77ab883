"""Multi-GPU plumbing of the frame-sharded path: no data-path collective exists -- every frame is a key
frame and every slice resets its contexts (Source/CLI/Global.cpp:959-960, FFV1_Slice.cpp:180-197,274-275) -- so the
process group is used for barriers and for reducing the timing only."""
from __future__ import annotations


def shard_frames(n_frames: int, rank: int, world: int, batch: int = 1) -> list[int]:
    """Frame indices rank `rank` encodes: batches of `batch` consecutive frames dealt round-robin (job.cpp does the same
    across the devices of one process: batch b -> device b mod n)."""
    out = []
    for b in range(rank, (n_frames + batch - 1) // batch, world):
        out += list(range(b * batch, min(n_frames, (b + 1) * batch)))
    return out


def max_over_ranks(dist, seconds: float, device) -> float:
    """Wall time of the slowest rank (the driver's contract for bench.py)."""
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
