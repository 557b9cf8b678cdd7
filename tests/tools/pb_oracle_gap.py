#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (uses oracle/; lives under tests/ for that reason).  How far is the deterministic PB-OSD restatement (oracle/ldpc_oracle.c orc_pb_osd: det_expf, float64 CDF
recurrence -- the form the HIP kernel is bit-exact to) from the LITERAL restatement of
PB_OSD/pb_testing.py:100-149 (NumPy exp, SciPy binom.cdf -- what the reference calls)?

CPU only.  For every SNR / order point: NMS-10 failures of synthetic frames, both restatements on each
frame, count the frames on which any of {num_teps, stop reason, comparisons, winner index, codeword}
differs.  The literal form is evaluated by `literal_pb` below: np_oracle.pb_osd_frame with its frontier kept
in arrays (same operations, same float types; checked against np_oracle.pb_osd_frame itself on the first
frames of every point).  Frames whose search runs past --cap TEPs are not compared (the literal Python loop is
quadratic in the frontier) and are reported as skipped.

    python tests/tools/pb_oracle_gap.py [--frames 2000] [--out profiles/r02/pb_oracle_gap.json]
"""
from __future__ import annotations

import argparse
import json
import math
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from oracle import c_oracle, np_oracle  # noqa: E402

F32 = np.float32
ALIST = os.path.join(ROOT, "short_ldpc_decoding_osd_amd", "data", "CCSDS_ldpc_n128_k64.alist")


def literal_pb(yp, Gp, order, snr_db, cap):
    """np_oracle.pb_osd_frame (literal NumPy/SciPy arithmetic) with the frontier in arrays."""
    import scipy.stats as stats
    yp = np.asarray(yp, dtype=F32)
    k, n = Gp.shape
    w = np.abs(yp)
    hard = np.where(yp > 0, 0, 1).astype(np.int64)
    c4 = F32(-4.0 * (1.0 / 10 ** (snr_db / 10)))
    q = np_oracle._sigmoid32(c4 * w)
    p1 = F32(np.add.reduce(q[k:], dtype=F32) / F32(n - k))
    pt = F32(np.add.reduce(q[:k], dtype=F32) / F32(k))
    lrb_mean = F32(np.add.reduce(w[k:], dtype=F32) / F32(n - k))
    spl = F32(1)
    for i in range(k):
        spl = F32(spl * (F32(1) - q[i]))
    niu = float(stats.binom.cdf(order, k, float(pt)))
    nmax = sum(math.comb(k, i) for i in range(order + 1))
    p_t_suc, p_t_pro = 0.99 * niu, 0.002 * math.sqrt((1 - niu) / nmax)
    cdfA = stats.binom.cdf(np.arange(65), n - k, float(p1))
    cdfH = stats.binom.cdf(np.arange(65), n - k, 0.5)

    def encode(support):
        mrb = hard[:k].copy()
        for p in support:
            mrb[p] ^= 1
        cw = mrb.dot(Gp) % 2
        disc = (cw + hard) % 2
        return cw, disc, np_oracle._weighted_distance_k(disc, w, k)

    def rsum(support):
        acc = F32(0)
        for p in sorted(support):
            acc = F32(acc + w[p])
        return acc

    best_cw, _, w_dmin = encode(())
    sums = np.empty(2 * cap + 8, dtype=F32)
    frontier = [(k - 1,)]
    sums[0] = rsum(frontier[0])
    nf = 1
    num_teps, best_index, comparisons, stop = nmax, 0, 0, 0
    for j in range(min(nmax - 1, cap)):
        mi = int(np.argmin(sums[:nf]))
        comparisons += 1 if nf == 1 else 2
        sel = frontier.pop(mi)
        rs = F32(sums[mi])
        sums[mi:nf - 1] = sums[mi + 1:nf]
        nf -= 1
        kids = []
        if sel[-1] < k - 1 and len(sel) < order:
            kids.append(sel + (k - 1,))
        if len(sel) > 1:
            if sel[-1] - sel[-2] > 1:
                kids.append(sel[:-1] + (sel[-1] - 1,))
        elif sel[-1] - 1 > -1:
            kids.append((sel[-1] - 1,))
        for c in kids:
            frontier.append(c)
            sums[nf] = rsum(c)
            nf += 1
        w1 = F32(np.exp(F32(c4 * rs)) * spl)
        w2 = F32(F32(1) - w1)
        bt = np.floor(F32(F32(w_dmin - rs) / lrb_mean))
        beta = 0 if not bt > 0 else (64 if bt > 64 else int(bt))
        bs = F32(0)
        bs = F32(bs + w1 * F32(cdfA[beta]))
        bs = F32(bs + w2 * F32(cdfH[beta]))
        if float(bs) < p_t_pro:
            stop, num_teps = 1, j + 1
            break
        cw, disc, wd = encode(sel)
        if wd < w_dmin:
            best_cw, w_dmin, best_index = cw, wd, j + 1
            ratio = F32(F32(F32(1) - w1) / w1)
            prod = F32(1)
            for i in range(k, n):
                prod = F32(prod * (F32(2) * q[i] if disc[i] else F32(2) * (F32(1) - q[i])))
            p_suc = F32(F32(1) / F32(F32(1) + F32(ratio / prod)))
            if p_suc > F32(p_t_suc):
                stop, num_teps = 2, j + 1
                break
    return dict(codeword=best_cw, metric=w_dmin, num_teps=num_teps, best_index=best_index,
                comparisons=comparisons, stop=stop)


def _work(args):
    y, cw, order, snr, cap, selfcheck = args
    code = np_oracle.Code(ALIST)
    res = c_oracle.pb_osd(code.G, y, cw, order, snr)
    out = []
    for j in range(y.shape[0]):
        pops = int(res["num_teps"][j]) if res["stop"][j] else int(res["num_teps"][j]) - 1   # no stop: N_max - 1 pops
        if pops > cap:
            out.append(("skipped", int(res["num_teps"][j])))
            continue
        yp, lp, Gp, perm, _ = np_oracle.swapped_info(y[j], cw[j], code.G)
        o = literal_pb(yp, Gp, order, snr, cap)
        if selfcheck and j < 3:
            o2 = np_oracle.pb_osd_frame(yp, lp, Gp, order, snr)
            assert all(o[k_] == o2[k_] for k_ in ("num_teps", "best_index", "comparisons", "stop")) and \
                np.array_equal(o["codeword"], o2["codeword"]), "literal_pb drifted from np_oracle.pb_osd_frame"
        cwo = np.empty(128, dtype=np.int64)
        cwo[perm] = o["codeword"]
        if o["stop"] == 0 and res["stop"][j] != 0 and pops < cap <= o["num_teps"] - 1:
            out.append(("literal_past_cap", int(res["num_teps"][j])))     # the literal loop ran into the cap: undecided
            continue
        same = (o["num_teps"] == res["num_teps"][j] and o["stop"] == res["stop"][j]
                and o["comparisons"] == res["comparisons"][j] and o["best_index"] == res["best_index"][j]
                and np.array_equal(cwo, res["codeword"][j]))
        same_cw = bool(np.array_equal(cwo, res["codeword"][j]))
        out.append(("same" if same else ("decision" if same_cw else "codeword"),
                    int(res["num_teps"][j]), int(o["num_teps"]), int(res["stop"][j]), int(o["stop"])))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=2000, help="NMS failures compared per point")
    ap.add_argument("--cap", type=int, default=6000, help="longest search (TEPs) the literal loop replays")
    ap.add_argument("--procs", type=int, default=max(1, (os.cpu_count() or 2) - 1))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02", "pb_oracle_gap.json"))
    args = ap.parse_args()
    code = np_oracle.Code(ALIST)
    points = [(snr, order) for snr in (1.0, 2.5, 3.5) for order in (2, 3)]
    fail_rate = {1.0: 0.75, 2.5: 0.25, 3.5: 0.05}
    report = []
    t00 = time.time()
    with mp.Pool(args.procs) as pool:
        for snr, order in points:
            rng = np.random.default_rng(int(snr * 100) + order)
            need = int(args.frames / fail_rate[snr] * 1.3) + 200
            y, cw = np_oracle.make_frames(code.G, snr, need, rng)
            soft = c_oracle.nms(code.H, y, 10, 0.669435)
            _, fail, _ = c_oracle.evaluate(code.H, soft, cw)
            idx = np.flatnonzero(fail)[:args.frames]
            parts = np.array_split(idx, max(1, len(idx) // 25))
            jobs = [(y[p], cw[p], order, snr, args.cap, i == 0) for i, p in enumerate(parts)]
            t0 = time.time()
            rows = [r for part in pool.imap_unordered(_work, jobs, chunksize=1) for r in part]
            kinds = [r[0] for r in rows]
            rec = dict(snr_db=snr, order=order, frames=len(rows), same=kinds.count("same"),
                       decision_differs=kinds.count("decision"), codeword_differs=kinds.count("codeword"),
                       skipped_longer_than_cap=kinds.count("skipped"), literal_past_cap=kinds.count("literal_past_cap"), cap=args.cap,
                       differing=[r for r in rows if r[0] in ("decision", "codeword")][:40],
                       seconds=round(time.time() - t0, 1))
            report.append(rec)
            print(json.dumps({k_: v for k_, v in rec.items() if k_ != "differing"}), flush=True)
            os.makedirs(os.path.dirname(args.out), exist_ok=True)
            with open(args.out, "w") as f:
                json.dump(dict(what="C det_expf/f64-recurrence PB-OSD vs literal NumPy exp / SciPy binom.cdf restatement "
                                    "(pb_testing.py:100-149), NMS-10 failures, CPU only",
                               total_seconds=round(time.time() - t00, 1), points=report), f, indent=1)


if __name__ == "__main__":
    main()
