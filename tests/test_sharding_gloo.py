"""CPU, world_size 2 over gloo: the N > 1 path (frame sharding + counter all-reduce)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from short_ldpc_decoding_osd_amd.sharding import allreduce_counters, rank_seed, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(total, rank, world)
    # stand-in for the per-rank decode: counters derived from the owned frame numbers
    g = torch.Generator().manual_seed(rank_seed(20241020, rank))
    _ = torch.rand(4, generator=g)
    frames = torch.arange(lo, hi, dtype=torch.int64)
    c = torch.tensor([hi - lo, int((frames % 4 == 0).sum()), int((frames % 7 == 0).sum()), 0,
                      int((frames % 4 == 0).sum()), int((frames % 4 == 0).sum()), int((frames % 64 == 0).sum()),
                      2081 * int((frames % 4 == 0).sum())], dtype=torch.int64)
    c = allreduce_counters(c)
    out.put((rank, lo, hi, c.tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [1000, 1001, 131072])
def test_world2_counters(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, c0), (r1, lo1, hi1, c1) = got
    assert (lo0, hi1) == (0, total) and hi0 == lo1 and abs((hi0 - lo0) - (hi1 - lo1)) <= 1
    assert c0 == c1
    frames = torch.arange(total)
    assert c0[0] == total and c0[1] == int((frames % 4 == 0).sum()) and c0[6] == int((frames % 64 == 0).sum())
    from short_ldpc_decoding_osd_amd.sharding import combine_fer
    f = combine_fer(c0)
    assert f["fer_product"] == pytest.approx(f["synd_fail_rate"] * f["fer_osd_given_fail"])
    assert f["mean_teps"] == 2081


def test_shard_range_edges():
    from short_ldpc_decoding_osd_amd.sharding import allreduce_counters, shard_range
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert shard_range(0, 0, 1) == (0, 0)
    with pytest.raises(ValueError):
        shard_range(10, 4, 4)
    t = torch.arange(8, dtype=torch.int64)
    assert torch.equal(allreduce_counters(t.clone()), t)       # not initialised: identity
    with pytest.raises(ValueError):
        allreduce_counters(torch.zeros(3))
