"""Sample-split rendering across ranks (one process per GPU, torch.distributed).

Frames are independent given their `time` value (compute.wgsl:85), so a job of F frames starting at
time0 is dealt round-robin: rank g of G renders time0 + g, time0 + g + G, ...  Every rank holds its
own fp32 sum; one sum-reduce (RCCL over xGMI on GPUs, gloo on CPU) to rank 0 gives the sum over all
F frames, which differs from the one-rank result only by fp32 summation order."""
from __future__ import annotations

from typing import Tuple


def frames_for_rank(rank: int, world: int, time0: int, n_frames: int) -> Tuple[int, int, int]:
    """-> (first time value, stride, frame count) of this rank's share of frames time0 .. time0+n-1."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    count = (n_frames - rank + world - 1) // world if n_frames > rank else 0
    return (time0 + rank) & 0xFFFFFFFF, world, count


def reduce_sum_to_root(accum, dst: int = 0, group=None):
    """In-place sum of the per-rank accumulation tensors onto rank `dst` (no-op for one rank)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return accum
