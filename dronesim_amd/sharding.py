"""Multi-GPU layout of a fleet: drones are independent on every live path of the reference
(Physics.PYB has no inter-drone term), so a fleet shards by contiguous index ranges, one process
per GPU, state never leaves its GPU and the data path needs no collective.  The only
communication is the timing/throughput reduction of a benchmark (or, with the optional
neighbour-downwash term, an all-gather of positions — see downwash.py).

Pure host logic (no device work): importable and testable on CPU with the gloo backend.
"""
from __future__ import annotations

from typing import Tuple


def shard_range(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """[begin, end) of the drones owned by `rank`: contiguous, sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(n_total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def rank_seed(noise_seed: int, rank: int) -> int:
    """Per-rank Threefry key: every rank numbers its drones from 0, so the key separates the streams.
    0 (noise off) stays 0."""
    if noise_seed == 0:
        return 0
    return (noise_seed + 0x9E3779B97F4A7C15 * rank) & 0xFFFFFFFFFFFFFFFF or 1


def reduce_step_times(dist, device, wall_s: float, dev_s: float) -> Tuple[float, float]:
    """MAX over ranks of (wall, device) seconds: a step is as slow as the slowest shard."""
    import torch
    t = torch.tensor([wall_s, dev_s], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0]), float(t[1])


def aggregate_throughput(drones_per_rank, steps: int, wall_max_s: float) -> float:
    """Whole-job drone-steps/s: all ranks' drones x steps over the slowest rank's time."""
    return float(sum(drones_per_rank)) * steps / wall_max_s
