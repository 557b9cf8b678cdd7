#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (uses oracle/; CPU only).  Where the first guess of PB-OSD's chunk bounds comes from, and how many
exact counts the bound picker of csrc/ldpc_osd_pb.hip (pb_pick_bound) needs.

    python tests/tools/pb_bound_model.py [--snr 2.5] [--frames 6000] [--sample 400]

For NMS decoding failures it sorts the reliability sums of all 43 744 TEPs of weight <= 3 and prints, for N = 256 ... 20000,
the quantiles of  T_N / (|y'_61| + |y'_62| + |y'_63|)  (T_N = the N-th smallest sum): the table `pb_bound_guess`
interpolates.  Then it replays the picker (first guess from the table, secant steps on log N over log T with the exact
counts, bisection as the fallback) over the chunk schedule of the kernels and prints counts per chunk and chunk sizes.
Exactness never depends on the bounds: this is a performance model only.
"""
from __future__ import annotations

import argparse
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import c_oracle, np_oracle  # noqa: E402
from short_ldpc_decoding_osd_amd import Code  # noqa: E402
from tests import pb_chunk_model as M  # noqa: E402

G = [(256, 0.80), (512, 0.89), (1024, 1.02), (2048, 1.14), (4096, 1.23), (8192, 1.33), (20000, 1.52), (43744, 2.2)]


def guess(n):
    n = max(n, 64)
    xs, ys = [math.log2(a) for a, _ in G], [b for _, b in G]
    if math.log2(n) <= xs[0]:
        return ys[0] * 2.0 ** ((math.log2(n) - xs[0]) / 6.0)
    return float(np.interp(math.log2(n), xs, ys))


def pick(w, s, lo, done, target, cap):
    count = lambda T: int(np.searchsorted(s, np.float32(T), side="right")) - done
    nall, inf = len(s), float("inf")
    m3 = float(np.float32(np.float32(w[61] + w[62]) + w[63]))
    want = done + target
    T = inf if nall - done <= cap else m3 * guess(want)
    if not T > lo:
        T = lo * 1.05 if lo > 0 else float(w[0])
    Tl, Th, tp, npt = lo, inf, lo, float(done)
    for it in range(41):
        c = count(T)
        if 0 < c <= cap and (T == inf or 5 * c >= 2 * target or it >= 2):
            return T, c, it + 1
        if T == inf:
            break
        if c == 0:
            Tl, Tn = T, T * 1.1
        else:
            if c > cap:
                Th = T
            tot, p = float(done + c), 6.0
            if tp > 0 and npt > 0 and tot != npt and T != tp:
                pe = (math.log2(tot) - math.log2(npt)) / (math.log2(T) - math.log2(tp))
                if 1.5 < pe < 20:
                    p = pe
            tp, npt = T, tot
            Tn = T * 2.0 ** ((math.log2(want) - math.log2(tot)) / p)
        if it >= 6 or not (Tl < Tn < Th):
            Tn = Tl + (Th - Tl) * 0.5 if Th < inf else T * 1.2
        if not (Tl < Tn < Th):
            break
        T = float(np.float32(Tn))
    return None, -1, 41


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--snr", type=float, default=2.5)
    ap.add_argument("--frames", type=int, default=6000)
    ap.add_argument("--sample", type=int, default=400)
    args = ap.parse_args()
    code = Code()
    rng = np.random.default_rng(2)
    y, cw = np_oracle.make_frames(code.G, args.snr, args.frames, rng)
    y = y.astype(np.float32)
    soft = c_oracle.nms(code.H, y, 10, 0.669435)
    _, fail, _ = c_oracle.evaluate(code.H, soft, cw)
    idx = np.flatnonzero(fail)[: args.sample]
    iu = np.triu_indices(64, 1)
    tri = np.array([(i, j, k) for i in range(62) for j in range(i + 1, 63) for k in range(j + 1, 64)])
    W, S = [], []
    for f in idx:
        w = np.asarray(M.frame_inputs(code.G, y[f])["w"])[:64].astype(np.float32)
        s = np.concatenate([w, (w[iu[0]] + w[iu[1]]).astype(np.float32),
                            ((w[tri[:, 0]] + w[tri[:, 1]]).astype(np.float32) + w[tri[:, 2]]).astype(np.float32)])
        s.sort()
        W.append(w)
        S.append(s)
    W, S = np.array(W), np.array(S)
    m3 = W[:, 61] + W[:, 62] + W[:, 63]
    print(f"{len(idx)} decoding failures at {args.snr} dB")
    for n in (256, 512, 1024, 2048, 4096, 8192, 20000):
        q = np.quantile(S[:, n - 1] / m3, [0.1, 0.5, 0.9])
        print(f"N = {n:6d}: T_N / m3  q10 {q[0]:.2f}  median {q[1]:.2f}  q90 {q[2]:.2f}   (table: {guess(n):.2f})")
    sched = [(768, 1024), (768, 1024)] + [(3072, 4096)] * 40
    per = {}
    for w, s in zip(W, S):
        lo, done, k = -1.0, 0, 0
        while done < len(s) and k < 6:
            T, c, it = pick(w, s, lo, done, *sched[k])
            if T is None:
                print("picker gave up (massive ties)")
                break
            per.setdefault(k, []).append((it, c))
            lo, done, k = T, done + c, k + 1
    for k, v in per.items():
        v = np.array(v)
        print(f"chunk {k}: counts per bound mean {v[:, 0].mean():.2f} max {v[:, 0].max()}; size q10/50/90 {np.quantile(v[:, 1], [0.1, 0.5, 0.9])}")


if __name__ == "__main__":
    main()
