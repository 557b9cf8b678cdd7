"""GPU box: python tests/tools/pb_long_fuzz.py [minutes] -- PB-OSD against the C oracle on fresh random batches until the time is up:
SNR 1.0-3.5 dB, orders 1-3, default schedule and random ones (budgets, chunk targets, tail rule), the default route and the
front-end-inside option; every count, stop reason, winner and metric exact.  (The long form of tests/test_gpu_fuzz.py.)"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import c_oracle, np_oracle
from short_ldpc_decoding_osd_amd import Code, _lib
from short_ldpc_decoding_osd_amd.runtime import Decoder
from tests.gpu_util import pack_np, to_dev, words_np
ALPHA0 = 0.669435
dec = Decoder(Code())
minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
t_end = time.time() + 60 * minutes
rng = np.random.default_rng(int(time.time()))
defaults = dec.pb_tuning()
total = rounds = 0
while time.time() < t_end:
    snr = float(rng.choice([1.0, 1.5, 2.0, 2.5, 3.0, 3.5]))
    order = int(rng.choice([1, 2, 3, 3, 3]))
    frames = int(rng.integers(300, 4000))
    seed = int(rng.integers(1 << 30))
    y, cw = np_oracle.make_frames(dec.code.G, snr, frames, np.random.default_rng(seed))
    q = 0.0
    if rng.random() < 0.3:
        q = float(rng.choice([16.0, 64.0, 256.0, 1024.0, 8192.0, 65536.0]))
        y = (np.round(y * q) / q).astype(np.float32)
    soft = c_oracle.nms(dec.code.H, y, 10, ALPHA0)
    _, fail, _ = c_oracle.evaluate(dec.code.H, soft, cw)
    idx = np.flatnonzero(fail)[:1500]
    if idx.size == 0:
        continue
    y, cw = y[idx], cw[idx]
    tuning = {}
    if rng.random() < 0.6:
        b = int(rng.choice([64, 300, 1000, 4096, 20000]))
        tuning = dict(budget_s=b, budget_m=b, budget=b, budget_l=b, budget_xl=b, t1=int(rng.integers(32, 833)), t2=int(rng.integers(32, 833)),
                      t3=int(rng.integers(256, 4097)), late_min=int(rng.choice([0, 4608])), late_pct=int(rng.choice([0, 20, 100, 100000])),
                      late_div=int(rng.choice([1, 4, 16, 64])))
    inside = bool(rng.random() < 0.4)
    path = None if inside or rng.random() < 0.85 else str(rng.choice(["block", "replay"]))
    dec.set_pb_tuning(**tuning) if tuning else dec.set_pb_tuning()
    ref = c_oracle.pb_osd(dec.code.G, y, cw, order, snr)
    aux = torch.zeros((y.shape[0], 4), dtype=torch.int32, device=dec.device)
    out = dec.osd_decode(to_dev(y, dec), order, params=dec.osd_params(order, _lib.OSD_PB, snr_db=snr, aux=aux, pb_front_inside=inside, pb_path=path))
    torch.cuda.synchronize()
    a = aux.cpu().numpy()
    ok = (np.array_equal(out["ntep"].cpu().numpy(), ref["num_teps"]) and np.array_equal(a[:, 3], ref["stop"]) and
          np.array_equal(a[:, 0], ref["comparisons"]) and np.array_equal(a[:, 1], ref["suc1"]) and np.array_equal(a[:, 2], ref["suc2"]) and
          np.array_equal(out["best"].cpu().numpy(), ref["best_index"]) and np.array_equal(words_np(out["cw"]), pack_np(ref["codeword"])) and
          np.array_equal(out["metric"].cpu().numpy(), ref["metric"]))
    if not ok:
        print("MISMATCH", dict(snr=snr, order=order, frames=frames, seed=seed, quant=q, tuning=tuning, inside=inside, path=path), flush=True)
        sys.exit(1)
    total += y.shape[0]; rounds += 1
    if rounds % 20 == 0:
        print(f"{rounds} rounds, {total} frame decodes exact", flush=True)
dec.set_pb_tuning()
print(f"done: {rounds} rounds, {total} frame decodes, all exact")
