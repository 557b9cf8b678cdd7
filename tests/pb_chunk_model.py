"""NumPy model of the batch-parallel PB-OSD engine of csrc/ldpc_osd_pb.hip -- TEST INFRASTRUCTURE.

The reference's PB-OSD (PB_OSD/pb_testing.py:100-149, optimal_tep_sequence :366-397) pops test error
patterns from a growing frontier list, "first minimum of the reliability sums in list order".  The kernel does
not replay that list.  It uses three facts (proved in DESIGN.md 3.4, checked here against the literal C oracle):

  1. every TEP has exactly one parent (extended child: e U {63}; adjacent child: largest index - 1), so the
     frontier never holds duplicates and the pop sequence visits every TEP of weight 1..order once;
  2. a child's float32 sum is >= its parent's, hence the pop sequence is the TEPs sorted by
     (sum, list slot), and  slot(t) < slot(u)  <=>  parent(t) popped before parent(u), or same parent and t is
     the extended child;
  3. the stopping rules depend on the visit order only through `best so far`, a prefix minimum.

So a chunk of the visit order = all TEPs whose sum lies in (lo, hi], sorted by sum, equal sums ordered by
rule 2 (recursively through the parents); costs of a chunk are evaluated in parallel and the sequential
rules are recovered with prefix scans.  This file states that algorithm with NumPy (same float32 operations
as the kernel and the C oracle) so that chunking, tie order, counters and stop logic can be checked on the CPU.
"""
from __future__ import annotations

import math

import numpy as np

from oracle import c_oracle

F32 = np.float32
DEGENERATE = "degenerate"   # frame left to the sequential kernel (a histogram bin or a tie group too large)


def pb_table(order, k=64):
    """TEPs of weight 1..order as (pos[N,3] ascending positions padded with -1, weight[N]) sorted by
    descending smallest position: the TEPs over positions >= a are the prefix of length plen[k - a]."""
    rows = []
    for p0 in range(k - 1, -1, -1):
        rows.append((p0, -1, -1, 1))
        if order >= 2:
            for p1 in range(p0 + 1, k):
                rows.append((p0, p1, -1, 2))
                if order >= 3:
                    for p2 in range(p1 + 1, k):
                        rows.append((p0, p1, p2, 3))
    t = np.array(rows, dtype=np.int64)
    plen = np.array([sum(math.comb(m, w) for w in range(1, order + 1)) for m in range(k + 1)], dtype=np.int64)
    return t[:, :3], t[:, 3], plen


def parent_of(t):
    """(parent support, child kind) of the support tuple t; kind 0 = extended child, 1 = adjacent child."""
    if t[-1] == 63:
        return (t[:-1], 0) if len(t) > 1 else (None, 0)      # {63} is the root
    return (t[:-1] + (t[-1] + 1,), 1)


def tsum(w, t):
    acc = F32(0)
    for i, p in enumerate(t):
        acc = w[p] if i == 0 else F32(acc + w[p])
    return acc


def visit_less(w, t, u):
    """t is popped before u (t != u)."""
    while True:
        st, su = tsum(w, t), tsum(w, u)
        if st != su:
            return st < su
        pt, kt = parent_of(t)
        pu, ku = parent_of(u)
        if pt is None:
            return True
        if pu is None:
            return False
        if pt == pu:
            return kt < ku
        t, u = pt, pu


def det_expf(x):
    return c_oracle.det_expf(np.asarray(x, dtype=F32))


def frame_inputs(G, y):
    """Primed-domain quantities of one frame from the C oracle's front end."""
    perm, Gp, _ = c_oracle.osd_front(G, y)
    yp = np.asarray(y, dtype=F32)[perm]
    w = np.abs(yp)
    hard = (~(yp > 0)).astype(np.uint64)
    hm = sum(int(hard[p]) << p for p in range(64))
    hp = sum(int(hard[64 + p]) << p for p in range(64))
    P = [sum(int(Gp[r, 64 + c] & 1) << c for c in range(64)) for r in range(64)]
    d0 = hp
    for r in range(64):
        if (hm >> r) & 1:
            d0 ^= P[r]
    return dict(w=w, P=P, d0=d0, hm=hm, hp=hp, perm=perm)


def pb_chunk_frame(fr, order, snr_db, cap=256, nbins=256, schedule=(48, 32, 16, 0), max_tie=16, budget=None, tables=None):
    """PB-OSD of one frame by sorted chunks.  Returns the C oracle's outputs (num_teps, best_index,
    comparisons, suc1, suc2, stop, metric, D, E), or DEGENERATE, or "budget" when `budget` pops did not stop."""
    w, P, d0 = fr["w"], fr["P"], fr["d0"]
    pos, wt, plen = tables if tables is not None else pb_table(order)
    N = len(wt)
    nmax = N + 1
    # ---- per-frame PB quantities, float conventions of oracle/ldpc_oracle.c orc_pb_osd
    c4 = F32(-4.0 * (1.0 / math.pow(10.0, float(snr_db) / 10.0)))
    q = (F32(1) / (F32(1) + det_expf(-(c4 * w)))).astype(F32)
    a1 = aw = at = F32(0)
    spl = F32(1)
    for p in range(64):
        a1 = F32(a1 + q[64 + p]); aw = F32(aw + w[64 + p]); at = F32(at + q[p]); spl = F32(spl * (F32(1) - q[p]))
    p1, lrb_mean, pt = F32(a1 / F32(64)), F32(aw / F32(64)), F32(at / F32(64))
    cdfA, cdfH = c_oracle.binom_cdf64(float(p1)), c_oracle.binom_cdf64(0.5)
    niu = c_oracle.binom_cdf64(float(pt))[order]
    p_t_suc, p_t_pro = 0.99 * niu, 0.002 * math.sqrt((1.0 - niu) / nmax)
    lut = np.zeros((8, 256), dtype=F32)
    for b in range(8):
        for t in range(8):
            for v in range(1 << t, 2 << t):
                lut[b][v] = F32(lut[b][v - (1 << t)] + w[64 + 8 * b + t])

    def cost_of(rs, D):
        acc = rs.astype(F32)
        for b in range(8):
            acc = (acc + lut[b][(D >> np.uint64(8 * b)) & np.uint64(0xFF)]).astype(F32)
        return acc

    # ---- all sums, ascending-position sequential float32
    p0, p1_, p2 = pos[:, 0], pos[:, 1], pos[:, 2]
    s = w[p0].astype(F32)
    s = np.where(wt >= 2, (s + w[np.maximum(p1_, 0)]).astype(F32), s)
    s = np.where(wt >= 3, (s + w[np.maximum(p2, 0)]).astype(F32), s)
    Pn = np.array(P, dtype=np.uint64)
    last = np.where(wt == 1, p0, np.where(wt == 2, p1_, p2))
    prev = np.where(wt == 2, p0, p1_)
    has1 = (last < 63) & (wt < order)
    has2 = np.where(wt > 1, last - prev > 1, last - 1 > -1)
    delta = has1.astype(np.int64) + has2.astype(np.int64) - 1
    wmax3 = F32(F32(w[0] + w[1]) + w[2])

    best = cost_of(np.array([0], dtype=F32), np.array([d0], dtype=np.uint64))[0]
    st = dict(j=0, nlive=1, cmp=0, suc1=0, suc2=0, stop=0, best=best, bestD=d0, bestE=0, bestidx=0)
    lo = F32(-1)
    for theta in [w[a] for a in schedule] + [F32(np.inf)]:
        while True:
            if np.isfinite(theta):
                le = np.flatnonzero(w[:64] <= theta)
                m = 64 - int(le[0]) if le.size else 0
            else:
                m = 64
            ncand = int(plen[m])
            sc = s[:ncand]
            sel = (sc > lo) & (sc <= theta)
            if not sel.any():
                break
            hi_val = F32(min(theta, wmax3))
            lo0 = F32(max(lo, F32(0)))
            span = F32(hi_val - lo0)
            if span > 0:
                scale = F32(F32(nbins) / span)
                bins = np.minimum(nbins - 1, (((sc - lo0).astype(F32) * scale).astype(F32)).astype(np.int64))
            else:
                bins = np.zeros(ncand, dtype=np.int64)
            hist = np.bincount(bins[sel], minlength=nbins)
            cum = np.cumsum(hist)
            if cum[-1] <= cap:
                bstar = nbins - 1
            else:
                ok = np.flatnonzero(cum <= cap)
                if ok.size == 0:
                    return DEGENERATE
                bstar = int(ok[-1])
                if cum[bstar] == 0:          # leading empty bins only, the first non-empty one is too large
                    return DEGENERATE
            ids = np.flatnonzero(sel & (bins <= bstar))
            # sort by (sum bits, id), then repair runs of equal sums with the list-order rule
            key = (s[ids].view(np.uint32).astype(np.uint64) << np.uint64(32)) | ids.astype(np.uint64)
            ids = ids[np.argsort(key, kind="stable")]
            ss = s[ids]
            i = 0
            while i < len(ids):
                e = i + 1
                while e < len(ids) and ss[e] == ss[i]:
                    e += 1
                if e - i > max_tie:
                    return DEGENERATE
                if e - i > 1:
                    grp = [int(t) for t in ids[i:e]]
                    tup = {t: tuple(int(x) for x in pos[t][:wt[t]]) for t in grp}
                    for a in range(1, len(grp)):      # insertion sort with the visit-order comparator
                        b = a
                        while b > 0 and visit_less(w, tup[grp[b]], tup[grp[b - 1]]):
                            grp[b], grp[b - 1] = grp[b - 1], grp[b]
                            b -= 1
                    ids[i:e] = grp
                i = e
            # ---- evaluate the chunk in parallel, recover the sequential rules with scans
            n = len(ids)
            rs = s[ids]
            D = np.full(n, d0, dtype=np.uint64)
            E = np.zeros(n, dtype=np.uint64)
            for col in range(3):
                pc = pos[ids, col]
                use = wt[ids] > col
                D = np.where(use, D ^ Pn[np.maximum(pc, 0)], D)
                E = np.where(use, E | (np.uint64(1) << np.maximum(pc, 0).astype(np.uint64)), E)
            cost = cost_of(rs, D)
            before = np.minimum.accumulate(np.concatenate([[st["best"]], cost]).astype(F32))[:-1]
            w1 = (det_expf(c4 * rs) * spl).astype(F32)
            w2 = (F32(1) - w1).astype(F32)
            bt = np.floor(((before - rs).astype(F32) / lrb_mean).astype(F32))
            beta = np.where(bt > 0, np.where(bt < 64, bt, 64), 0).astype(np.int64)
            bs = (F32(0) + (w1 * cdfA[beta].astype(F32)).astype(F32)).astype(F32)
            bs = (bs + (w2 * cdfH[beta].astype(F32)).astype(F32)).astype(F32)
            stop1 = bs.astype(np.float64) < p_t_pro
            newbest = cost < before
            stop2 = np.zeros(n, dtype=bool)
            for x in np.flatnonzero(newbest):
                ratio = F32(F32(F32(1) - w1[x]) / w1[x])
                prod = F32(1)
                for p in range(64):
                    qp = q[64 + p]
                    prod = F32(prod * (F32(F32(2) * qp) if (int(D[x]) >> p) & 1 else F32(F32(2) * F32(F32(1) - qp))))
                p_suc = F32(F32(1) / F32(F32(1) + F32(ratio / prod)))
                stop2[x] = float(p_suc) > p_t_suc
            nlive_before = st["nlive"] + np.concatenate([[0], np.cumsum(delta[ids])[:-1]])
            anystop = np.flatnonzero(stop1 | stop2)
            upto = int(anystop[0]) if anystop.size else n - 1          # last popped index of this chunk
            st["cmp"] += int(np.where(nlive_before[:upto + 1] == 1, 1, 2).sum())
            if anystop.size and stop1[upto]:
                nevald, reason = upto, 1
            elif anystop.size:
                nevald, reason = upto + 1, 2
            else:
                nevald, reason = n, 0
            st["suc1"] += nevald
            nb = np.flatnonzero(newbest[:nevald])
            st["suc2"] += int(nb.size)
            if nb.size:
                x = int(nb[-1])
                st.update(best=cost[x], bestD=int(D[x]), bestE=int(E[x]), bestidx=st["j"] + x + 1)
            if reason:
                st.update(stop=reason, ntep=st["j"] + upto + 1)
                return st
            st["j"] += n
            st["nlive"] = int(st["nlive"] + delta[ids].sum())
            if budget is not None and st["j"] >= budget:
                return "budget"
            if bstar == nbins - 1:
                break
            lo = rs.max()
        if np.isfinite(theta):
            lo = F32(max(lo, theta))
    assert st["j"] == N and st["nlive"] == 0, (st["j"], N, st["nlive"])
    st["ntep"] = nmax
    return st
