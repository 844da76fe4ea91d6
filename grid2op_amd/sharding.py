"""Static sharding of the lane batch over the GPUs of one node.

Lanes (environment copies / N-1 contingencies) never communicate, so the batch is cut into contiguous blocks, one
per rank (one process per GPU); there is NO collective on the data path.  ``torch.distributed`` (RCCL on the GPU
box, gloo in the CPU tests) is only used for the barrier and the max-over-ranks timing that ``bench.py`` reports.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

__all__ = ["lane_range", "synthetic_lane_inputs", "max_over_ranks", "sum_over_ranks"]


def lane_range(total_lanes: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block ``[lane0, lane0+n)`` of global lane ids owned by ``rank`` (sizes differ by at most 1)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(int(total_lanes), int(world))
    n = base + (1 if rank < rem else 0)
    lane0 = rank * base + min(rank, rem)
    return lane0, n


def synthetic_lane_inputs(n_load: int, T: int, lane_ids) -> Tuple[np.ndarray, np.ndarray]:
    """The synthetic workload of SURVEY.md 8(d) cfg 2, a pure function of the GLOBAL lane id (so that any
    sharding reproduces the same global batch): lane k reads chronics row ``(t + 7k) mod T`` and scales its loads by
    ``1 + 0.05 N(0,1)`` drawn from ``numpy.random.default_rng(k)``."""
    lane_ids = np.asarray(lane_ids, dtype=np.int64)
    offsets = ((7 * lane_ids) % T).astype(np.int32)
    scale = np.empty((lane_ids.size, 2 * n_load), dtype=np.float32)
    for i, k in enumerate(lane_ids):
        scale[i] = 1.0 + 0.05 * np.random.default_rng(int(k)).standard_normal(2 * n_load)
    return offsets, scale


def max_over_ranks(value: float, dist=None, device=None) -> float:
    """MAX-reduce a host scalar over the ranks (identity when not distributed)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, dist=None, device=None) -> float:
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
