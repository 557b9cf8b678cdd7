"""Multi-GPU: frames are independent, so a batch shards across ranks with no data-path
collective; the only exchange is one sum all-reduce of the error counters per measurement
(the reference itself multiplies stage FERs by hand, recipe.txt:18).
One process per GPU; backend "nccl" is RCCL over xGMI on ROCm, "gloo" is used by CPU tests.
"""
from __future__ import annotations

import torch


def shard_range(total_frames: int, rank: int, world: int):
    """Contiguous slice [lo, hi) of ``total_frames`` owned by ``rank`` (sizes differ by <= 1)."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError(f"rank {rank} of {world}")
    base, extra = divmod(int(total_frames), world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def rank_seed(base_seed: int, rank: int) -> int:
    """Per-rank generator seed (seed = 20241020 + rank)."""
    return int(base_seed) + int(rank)


def allreduce_counters(counters: torch.Tensor) -> torch.Tensor:
    """Sum int64 counters over all ranks (identity when torch.distributed is not initialised).
    A <= 64-byte message: latency-bound, one call per SNR point."""
    import torch.distributed as dist

    if counters.dtype != torch.int64:
        raise ValueError("counters must be int64")
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(counters, op=dist.ReduceOp.SUM)
    return counters


def combine_fer(counters) -> dict:
    """Readable rates from the 8 summed counters {frames, frame_err, bit_err, undetected,
    synd_fail, osd_frames, osd_wrong, teps}: both factors and their product, as the recipe
    multiplies them (Training and Testing recipe.txt:18)."""
    c = [int(x) for x in counters]
    frames = max(c[0], 1)
    out = {"frames": c[0], "fer_nms": c[1] / frames, "synd_fail_rate": c[4] / frames, "undetected": c[3]}
    if len(c) >= 8 and c[5] > 0:
        out["fer_osd_given_fail"] = c[6] / c[5]
        out["fer_product"] = out["synd_fail_rate"] * out["fer_osd_given_fail"]
        out["fer_end_to_end"] = (c[6] + c[3]) / frames
        out["mean_teps"] = c[7] / c[5]
    return out
