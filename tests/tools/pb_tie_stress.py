"""GPU box: python tests/tools/pb_tie_stress.py -- PB-OSD on finely to coarsely quantised channel values (equal sums everywhere) under three
hand-over schedules, every result against the C oracle (8640 frame decodes; the long form of test_pb_ties_under_schedules)."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import c_oracle, np_oracle
from short_ldpc_decoding_osd_amd import Code
from short_ldpc_decoding_osd_amd.runtime import Decoder
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tests.test_gpu_osd_pb import _check, ALPHA0
dec = Decoder(Code())
n = 0
for tuning in (dict(budget_s=64, budget_m=64, budget=64, budget_l=64, budget_xl=64), dict(budget_s=700, budget_m=700, budget=700), dict(late_pct=100000, late_div=64, late_min=0)):
    prev = dec.set_pb_tuning(**tuning)
    for quant in (64.0, 256.0, 2048.0, 65536.0):
        for seed in (1, 2):
            for snr, order in ((1.0, 3), (2.0, 3), (1.5, 2)):
                rng = np.random.default_rng(int(quant) * 10 + seed)
                y, cw = np_oracle.make_frames(dec.code.G, snr, 500, rng)
                y = (np.round(y * quant) / quant).astype(np.float32)
                soft = c_oracle.nms(dec.code.H, y, 10, ALPHA0)
                _, fail, _ = c_oracle.evaluate(dec.code.H, soft, cw)
                idx = np.flatnonzero(fail)[:120]
                ref = _check(dec, y[idx], cw[idx], order, snr, None)
                n += len(idx)
    dec.set_pb_tuning(**prev)
    print("tuning", tuning, "ok", n, flush=True)
print("all exact:", n, "frame decodes")
