"""CPU, world_size 2 over gloo: the N > 1 path (frame sharding + counter all-reduce).  There is no CPU decode path
in the product, so the ranks decode their shard of REAL frames with the C oracle (the checker) and the reduced
counters are compared with a single-process decode of the whole batch."""
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


def _oracle_counters(y, cw):
    """{frames, frame_err, bit_err, undetected, synd_fail, osd_frames, osd_wrong, teps} of NMS-10 + OSD-1 by the C oracle."""
    import numpy as np
    from oracle import c_oracle, np_oracle
    code = np_oracle.Code(os.path.join(ROOT, "short_ldpc_decoding_osd_amd", "data", "CCSDS_ldpc_n128_k64.alist"))
    soft = c_oracle.nms(code.H, y, 10, 0.669435)
    hard, fail, cnt = c_oracle.evaluate(code.H, soft, cw)
    idx = np.flatnonzero(fail)
    wrong = int((~c_oracle.conv_osd(code.G, y[idx], cw[idx], 1)["correct"]).sum()) if idx.size else 0
    return [y.shape[0], cnt["frame_err"], cnt["bit_err"], cnt["undetected"], cnt["synd_fail"], idx.size, wrong, 65 * idx.size]


def _decode_worker(rank, world, port, total, out):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch.distributed as dist
    from oracle import np_oracle
    from short_ldpc_decoding_osd_amd.sharding import allreduce_counters, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    code = np_oracle.Code(os.path.join(ROOT, "short_ldpc_decoding_osd_amd", "data", "CCSDS_ldpc_n128_k64.alist"))
    y, cw = np_oracle.make_frames(code.G, 2.5, total, np.random.default_rng(99))     # the same global batch on every rank
    lo, hi = shard_range(total, rank, world)
    c = allreduce_counters(torch.tensor(_oracle_counters(y[lo:hi], cw[lo:hi]), dtype=torch.int64))
    out.put((rank, c.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_decodes_real_frames():
    """Each rank decodes its contiguous shard; the all-reduced counters equal the single-process decode."""
    import numpy as np
    from oracle import np_oracle
    total = 3001
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_decode_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    code = np_oracle.Code(os.path.join(ROOT, "short_ldpc_decoding_osd_amd", "data", "CCSDS_ldpc_n128_k64.alist"))
    y, cw = np_oracle.make_frames(code.G, 2.5, total, np.random.default_rng(99))
    want = [int(v) for v in _oracle_counters(y, cw)]
    assert got[0][1] == want and got[1][1] == want
    assert want[4] > 500 and want[6] > 0          # the batch really has NMS failures and OSD-1 errors


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus N` without a launcher spawns its own ranks -- or fails loudly, never a 1-GPU number."""
    import subprocess
    have = torch.cuda.device_count()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(have + 2), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK")})
    assert p.returncode != 0 and "GPU(s) visible" in p.stderr and not p.stdout.strip()
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr and not p.stdout.strip()


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


# ---------------------------------------------------------------------------------------------------------
# the reference's stop rule in the sharded path (sharding.sweep_point): one counter all-reduce per macro-batch
# ---------------------------------------------------------------------------------------------------------
def _stop_worker(rank, world, port, total, batch, stop_errors, out):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch.distributed as dist
    from oracle import np_oracle
    from short_ldpc_decoding_osd_amd.sharding import shard_range, sweep_point
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    code = np_oracle.Code(os.path.join(ROOT, "short_ldpc_decoding_osd_amd", "data", "CCSDS_ldpc_n128_k64.alist"))
    y, cw = np_oracle.make_frames(code.G, 2.5, total, np.random.default_rng(5))      # the same global batch on every rank
    lo, hi = shard_range(total, rank, world)
    pos = [lo]

    def decode_batch(B):
        s = pos[0]
        pos[0] += B
        return torch.tensor(_oracle_counters(y[s:s + B], cw[s:s + B]), dtype=torch.int64)

    max_batches = -(-shard_range(total, 0, world)[1] // batch)
    red, ran = sweep_point(decode_batch, hi - lo, batch, max_batches, stop_errors, with_osd=True)
    out.put((rank, ran, red.tolist(), pos[0] - lo))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("stop_errors", [0, 12])
def test_world2_stop_rule(stop_errors):
    """Both ranks leave an SNR point on the SAME macro-batch, with the totals a single process gets for the frames that were
    decoded (pb_testing.py:174 / ldpc_128_testing.py:130 in the sharded path)."""
    import numpy as np
    from oracle import np_oracle
    from short_ldpc_decoding_osd_amd.sharding import end_to_end_errors, shard_range
    total, batch, world = 2401, 300, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_stop_worker, args=(r, world, port, total, batch, stop_errors, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, ran0, red0, used0), (_, ran1, red1, used1) = got
    assert ran0 == ran1 and red0 == red1                     # same macro-batch, same sums on both ranks
    # single-process replay of exactly the frames the ranks decoded
    code = np_oracle.Code(os.path.join(ROOT, "short_ldpc_decoding_osd_amd", "data", "CCSDS_ldpc_n128_k64.alist"))
    y, cw = np_oracle.make_frames(code.G, 2.5, total, np.random.default_rng(5))
    want = np.zeros(8, dtype=np.int64)
    for r, used in ((0, used0), (1, used1)):
        lo, _ = shard_range(total, r, world)
        want += np.array(_oracle_counters(y[lo:lo + used], cw[lo:lo + used]), dtype=np.int64)
    assert red0 == want.tolist()
    if stop_errors:
        assert end_to_end_errors(red0, True) >= stop_errors and red0[0] < total     # stopped early ...
        # ... and not a macro-batch too late: one macro-batch less would have been below the threshold
        less = np.zeros(8, dtype=np.int64)
        for r, used in ((0, used0), (1, used1)):
            lo, _ = shard_range(total, r, world)
            u = max(used - batch, 0)
            less += np.array(_oracle_counters(y[lo:lo + u], cw[lo:lo + u]), dtype=np.int64) if u else 0
        assert end_to_end_errors(less.tolist(), True) < stop_errors
    else:
        assert red0[0] == total


def _empty_shard_worker(rank, world, port, stop_errors, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from short_ldpc_decoding_osd_amd.sharding import shard_range, sweep_point
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total = 1                                              # fewer frames than ranks: rank 1's shard is empty
    lo, hi = shard_range(total, rank, world)
    calls = []

    def decode_batch(B):
        calls.append(B)
        return torch.tensor([B, 1, 3, 0, 1, 1, 1, 7], dtype=torch.int64)

    red, ran = sweep_point(decode_batch, hi - lo, 4, 1, stop_errors, with_osd=True, device=torch.device("cpu"))
    out.put((rank, ran, red.tolist(), calls, str(red.device)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("stop_errors", [0, 1])
def test_world2_rank_with_an_empty_shard(stop_errors):
    """--frames smaller than the world size: the rank without frames never calls decode_batch, contributes zero counters
    allocated on the device it was given (under nccl a CPU tensor cannot be reduced) and leaves with the others."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_empty_shard_worker, args=(r, 2, port, stop_errors, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, ran0, red0, calls0, dev0), (_, ran1, red1, calls1, dev1) = got
    assert calls0 == [1] and calls1 == []
    assert ran0 == ran1 == 1 and red0 == red1 == [1, 1, 3, 0, 1, 1, 1, 7]
    assert dev0 == dev1 == "cpu"


# ---------------------------------------------------------------------------------------------------------
# bench.py's own rank launcher (no GPU call in the parent; dummy workers)
# ---------------------------------------------------------------------------------------------------------
def _run_spawn(worker_code, timeout=20.0):
    import subprocess
    script = (
        "import sys, types; sys.path.insert(0, %r); import bench\n"
        "args = types.SimpleNamespace(gpus=3, rank_timeout=%r)\n"
        "bench.spawn_ranks(args, [], worker=[sys.executable, '-c', %r], have=3, poll=0.05)\n" % (ROOT, timeout, worker_code))
    return subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=120)


def test_spawn_ranks_relays_rank0_line():
    p = _run_spawn("import os, json; r = int(os.environ['RANK']); assert os.environ['WORLD_SIZE'] == '3' and os.environ['MASTER_ADDR'] == '127.0.0.1'; "
                   "print(json.dumps({'rank': r})) if r == 0 else None")
    assert p.returncode == 0 and p.stdout.strip() == '{"rank": 0}', (p.stdout, p.stderr)


def test_spawn_ranks_kills_the_others_when_one_rank_dies():
    """Rank 2 dies; ranks 0 and 1 would wait for ever (a barrier nobody completes): the launcher must notice, kill them and fail."""
    import time
    t0 = time.monotonic()
    p = _run_spawn("import os, sys, time; r = int(os.environ['RANK']); sys.exit(7) if r == 2 else time.sleep(600)")
    assert p.returncode != 0 and "rank 2 exited with code 7" in p.stderr and not p.stdout.strip()
    assert time.monotonic() - t0 < 60


def test_spawn_ranks_overall_timeout():
    p = _run_spawn("import time; time.sleep(600)", timeout=1.0)
    assert p.returncode != 0 and "still running" in p.stderr and not p.stdout.strip()
