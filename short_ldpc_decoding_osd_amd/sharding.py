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


def end_to_end_errors(counters, with_osd: bool) -> int:
    """Frames still wrong at the end of the path: OSD failures + NMS errors the syndrome did not flag (they are never
    post-processed, ms_test.py:51); without an OSD stage the NMS frame errors."""
    c = [int(x) for x in counters]
    return c[6] + c[3] if with_osd else c[1]


def sweep_point(decode_batch, frames_mine: int, batch: int, max_batches: int, stop_errors: int = 0, with_osd: bool = True,
                device=None):
    """The macro-batch loop of ONE SNR point on one rank, with the reference's stop rule.

    ``decode_batch(B)`` decodes the next B frames of this rank's shard and returns their int64[8] counters
    {frames, frame_err, bit_err, undetected, synd_fail, osd_frames, osd_wrong, teps} (a tensor on the rank's device).
    Every rank runs the SAME number of iterations (``max_batches`` = the largest shard's batch count; a rank whose shard is
    used up contributes zeros), so the collectives line up.

    stop_errors = 0: every frame is decoded and ONE all-reduce closes the point (the default of scripts/snr_sweep.py).
    stop_errors = N: the reference stops a point once enough errors were seen -- ldpc_128_testing.py:130 (accumulated
    frame errors > 40000 / batch), pb_testing.py:174 / fs_testing.py:199 (fail_sum >= termination_num_threshlod = 100,
    PB_OSD/globalmap.py:43).  Here: after every macro-batch the running totals are all-reduced (64 bytes) and
    the point ends as soon as the end-to-end frame errors of ALL ranks together reach N.  All ranks see the same sums, so
    they leave on the same macro-batch, and the returned totals are those of the frames actually decoded.

    ``device``: where this rank's counters live (the rank's GPU under the ``nccl`` backend -- RCCL cannot reduce a CPU tensor,
    and a rank whose shard is EMPTY never gets a tensor from ``decode_batch`` to copy the device from; default: CPU, which
    is what the gloo tests use).

    Returns (reduced int64[8] totals, macro-batches run)."""
    def zeros():
        return torch.zeros(8, dtype=torch.int64, device=device)

    total = None
    done = 0
    ran = 0
    for _ in range(max(1, int(max_batches))):
        B = min(int(batch), int(frames_mine) - done)
        c = decode_batch(B) if B > 0 else None
        if c is not None:
            total = c.clone() if total is None else total + c
        done += max(B, 0)
        ran += 1
        if stop_errors > 0:
            cur = total.clone() if total is not None else zeros()
            red = allreduce_counters(cur)
            if end_to_end_errors(red.tolist(), with_osd) >= stop_errors:
                return red, ran
    if stop_errors > 0:
        return red, ran
    return allreduce_counters(total if total is not None else zeros()), ran


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
