"""-m gpu: the RCCL leg on hardware.  A one-GPU box can only hold one rank, so the test initialises the `nccl`
(= RCCL) backend with world_size 1 in a CHILD process, pushes the device counter tensor of a real
BatchPipeline.run() through sharding.allreduce_counters and checks it against the same run without
torch.distributed; and bench.py's own rank launcher is exercised with `--gpus 1` under a launcher environment
(the path a multi-GPU run takes: init_process_group, barriers, MAX-reduced time) and refused for `--gpus 2`."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys, json
sys.path.insert(0, %(root)r)
import numpy as np, torch
import torch.distributed as dist
from short_ldpc_decoding_osd_amd import Code
from short_ldpc_decoding_osd_amd.runtime import Decoder
from short_ldpc_decoding_osd_amd.pipeline import BatchPipeline
from short_ldpc_decoding_osd_amd.sharding import allreduce_counters, combine_fer
from oracle import np_oracle
from tests.gpu_util import pack_np, to_dev
torch.cuda.set_device(0)
dec = Decoder(Code(), 0)
y, cw = np_oracle.make_frames(dec.code.G, 2.5, 20000, np.random.default_rng(3))
pipe = BatchPipeline(dec, 20000, 10, 0.669435, osd_order=2).bind(to_dev(y, dec), to_dev(pack_np(cw).view(np.int64), dec))
pipe.run(); torch.cuda.synchronize()
plain = pipe.counters().cpu().tolist()
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
pipe.reset_counters(); pipe.run()
red = allreduce_counters(pipe.counters().clone())          # device tensor through RCCL
t = torch.tensor([1.5], dtype=torch.float64, device=dec.device)
dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier(); torch.cuda.synchronize()
print(json.dumps(dict(plain=plain, reduced=red.cpu().tolist(), world=dist.get_world_size(), backend=dist.get_backend(),
                      tmax=float(t.item()), fer=combine_fer(red.cpu().numpy())["fer_end_to_end"])))
dist.destroy_process_group()
"""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra)
    return env


def test_rccl_allreduce_of_real_counters():
    env = _env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    p = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT)], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["backend"] == "nccl" and out["world"] == 1
    assert out["reduced"] == out["plain"] and out["plain"][0] == 20000 and out["plain"][5] > 3000
    assert out["tmax"] == 1.5 and 0.005 < out["fer"] < 0.03


def test_bench_launcher_path_and_refusal():
    # under a launcher environment (what torchrun sets): the distributed branch with one rank
    env = _env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), LDPC_BENCH_FORCE_DIST="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "8", "--warmup", "2", "--batch", "16384",
                        "--no-cpu-baseline", "--no-overlap-pass"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["rccl_ranks"] == 1 and line["config"]["global_frames_per_step"] == 16384
    assert line["config"]["distinct_frames"] == 4 * 16384 and line["roofline"]["avg_launch_ms"] > 0
    # plain start with more GPUs than the box has: a clear error, never an n_gpus: 1 line
    import torch
    have = torch.cuda.device_count()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(have + 1), "--steps", "2"], capture_output=True,
                       text=True, timeout=300, env=_env())
    assert p.returncode != 0 and "GPU(s) visible" in p.stderr and not p.stdout.strip()
