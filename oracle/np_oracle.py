"""NumPy oracle -- TEST INFRASTRUCTURE ONLY.

A CPU restatement of the reference's NMS + OSD hot path, written as close to the
reference's *dense / per-frame* formulation as NumPy allows, so that it can be read
side by side with the reference.  Nothing in the product package
(``short_ldpc_decoding_osd_amd/``) may import this module; only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do.

Pinning status (see DESIGN.md "Oracle"):
  * alist loader, G construction and the GF(2) elimination rule are pinned against the
    reference's own importable NumPy module ``fill_matrix_info.py`` through the golden
    fixtures in ``tests/golden`` (made by ``oracle/gen_golden.py``).
  * everything that the reference executes through TensorFlow (NMS float math, argsort,
    reduce_sum order) is a restatement of the cited lines; TensorFlow is absent from the
    build image, the reference ships no test vectors, so those float results are
    **parity unpinned** (GF(2)/integer parts remain pinned through the shared GE rule).

All file:line citations are relative to /root/reference/LDPC_128/.
"""
from __future__ import annotations

import itertools
import math

import numpy as np

F32 = np.float32

# --------------------------------------------------------------------------------------
# code definition:  Ldpc_128_testing/fill_matrix_info.py
# --------------------------------------------------------------------------------------


def load_alist(path):
    """alist text -> dense H[m,n] int64.  (fill_matrix_info.py:70-104)

    Line 1 ``n m``; line 2 max degrees; lines 3-4 degree lists (ignored by the
    reference too); then n lines of 1-based check ids per variable; tokens ``'0'`` and
    ``''`` are padding (:96).  The m check lines that follow are parsed by the reference
    but never used for H, so they are not read here.
    """
    with open(path, "rt") as fh:
        rows = [ln.rstrip("\n").split(" ") for ln in fh]
    n, m = (int(t) for t in rows[0][:2])
    max_var_deg, max_chk_deg = (int(t) for t in rows[1][:2])
    H = np.zeros((m, n), dtype=np.int64)
    for v in range(n):
        for tok in rows[4 + v]:
            if tok not in ("0", ""):
                H[int(tok) - 1, v] = 1
    return H, max_chk_deg


def gf2_eliminate(M):
    """Gauss-Jordan over GF(2) with the reference's pivoting rule.

    Restates ``full_gf2elim`` (PB_OSD/pb_testing.py:231-266), which is textually the same
    routine as ``Code.gf2elim`` (fill_matrix_info.py:7-42).  Works on a copy; returns the
    reduced matrix and the list of recorded column exchanges ``(j, col)``.

      for i = j = 0.. :  pivot = first row >= i with a 1 in column j          (:238-245)
                         none -> if row i is all-zero from j on: drop the row (:247-250)
                                 else swap column j with the first column >= j
                                 holding a 1 *in row i*, and record it        (:251-256)
                         clear column j in every other row (XOR of columns j:) (:258-262)
    """
    M = np.array(M, dtype=np.int64, copy=True)
    swaps = []
    i = j = 0
    m, n = M.shape
    while i < m and j < n:
        below = np.flatnonzero(M[i:, j])
        if below.size:
            r = i + int(below[0])
            if r != i:
                M[[i, r]] = M[[r, i]]
        else:
            right = np.flatnonzero(M[i, j:])
            if right.size == 0:
                M = np.delete(M, i, axis=0)
                m -= 1
                continue
            c = j + int(right[0])
            M[:, [j, c]] = M[:, [c, j]]
            swaps.append((j, c))
        hit = M[:, j].astype(bool)
        hit[i] = False
        M[hit, j:] ^= M[i, j:]
        i += 1
        j += 1
    return M, swaps


def generator_from_H(H):
    """Systematic-form generator from H.  (fill_matrix_info.py:44-69)

    GE gives ``[I | H2]``; ``G = [H2^T | I]``; the recorded column exchanges are undone in
    reverse order; ``H G^T = 0`` is asserted (the reference only prints, :63-68).
    """
    R, swaps = gf2_eliminate(H)
    m = R.shape[0]
    n = R.shape[1]
    G = np.concatenate([R[:, m:].T, np.eye(n - m, dtype=np.int64)], axis=1)
    for a, b in reversed(swaps):
        G[:, [a, b]] = G[:, [b, a]]
    assert not (H.dot(G.T) % 2).any(), "H G^T != 0"
    return G


class Code:
    """Attribute-compatible with the reference's ``Code`` (fill_matrix_info.py:3-129)."""

    def __init__(self, path):
        self.H, self.max_chk_degree = load_alist(path)
        self.check_matrix_row, self.check_matrix_column = self.H.shape
        self.G = generator_from_H(self.H)
        self.k = self.G.shape[0]


# --------------------------------------------------------------------------------------
# synthetic test frames:  Testing_data_gen_128/data_generating.py:13-51
# --------------------------------------------------------------------------------------


def snr_to_sigma(snr_db, k, n):
    """sigma = sqrt(1 / (2 R 10^(snr/10)))   (data_generating.py:17)"""
    return math.sqrt(1.0 / (2.0 * (float(k) / float(n)) * 10.0 ** (snr_db / 10.0)))


def make_frames(G, snr_db, frames, rng):
    """AWGN test frames: BPSK 0->+1, unit mean, no LLR scaling (:40-51).

    The reference draws from the unseeded global NumPy RNG (:10); here a Generator is
    passed in.  Returns (y float32 [F,n], codewords int64 [F,n]).
    """
    k, n = G.shape
    sigma = snr_to_sigma(snr_db, k, n)
    chan = rng.normal(1.0, sigma, size=(frames, n))
    msg = rng.integers(0, 2, size=(frames, k))
    cw = msg.dot(G) % 2
    y = np.where(cw == 0, chan, -chan)
    return y.astype(F32), cw.astype(np.int64)


# --------------------------------------------------------------------------------------
# NMS belief propagation, dense mirror:  Ldpc_128_testing/ms_test.py:99-242
# --------------------------------------------------------------------------------------


def softplus(x):
    return F32(np.log1p(np.exp(np.float64(x))))


def nms_dense(y, H, T, alpha, w_in=1.0, w_out=1.0):
    """Op-for-op dense mirror of ``Decoder_Layer`` for NMS-1/2/3 (float32 throughout).

    y [B,n] f32; alpha = effective check normaliser (softplus of the stored weight,
    ms_test.py:207-208), scalar or length-T; w_in / w_out = effective bit weights of
    NMS-2/3 (:127-131, :222-225; 1.0 for NMS-1).  Returns the list
    ``[y, out_1, ..., out_T]`` (:106-109, :228).
    """
    y = np.asarray(y, dtype=F32)
    B, n = y.shape
    m = H.shape[0]
    Hf = H.astype(F32)
    alpha = np.broadcast_to(np.asarray(alpha, dtype=F32), (T,))
    w_in = F32(w_in)
    w_out = F32(w_out)
    cv = np.zeros((B, m, n), dtype=F32)                                # :117
    outs = [y]
    back = np.where(H == 0, F32(-1e30) - F32(1), F32(0)).astype(F32)   # :193
    for it in range(T):
        # compute_vc :124-137
        tot = cv.sum(axis=1, dtype=F32) + y * w_in
        vc = tot[:, None, :] * Hf - cv
        # compute_cv2 signs :183-191
        sgn = np.sign((F32(1) - Hf)[None] + vc).astype(F32)
        rowprod = np.prod(sgn, axis=2, keepdims=True, dtype=F32)
        out_sign = (rowprod * Hf) * sgn
        # magnitudes :193-206
        a = np.clip(np.abs(vc), F32(0), F32(1e30))
        decision = -np.abs(a) + back[None]
        part = -np.sort(-decision, axis=2)[:, :, :2]                    # top_k(k=2) values
        m1 = (-part[:, :, 0:1]) * Hf
        m2 = (-part[:, :, 1:2]) * Hf
        mag = np.where(a > m1, m1, m2)
        cv = (alpha[it] * mag * out_sign).astype(F32)                   # :207-209
        # marginalize :220-228
        outs.append((cv.sum(axis=1, dtype=F32) + w_out * y).astype(F32))
    return outs


def nms_sparse(y, H, T, alpha, w_in=1.0, w_out=1.0):
    """Edge-list restatement of the same math (Appendix A.1 of SURVEY.md).

    Per-variable sums run over the variable's checks in ascending check index and add the
    channel value last -- the order a sequential dense ``reduce_sum(axis=1)`` (ms_test.py:132)
    produces, since adding the zero entries of non-edges is exact.  This is the order the
    C oracle and the HIP kernels follow bit-for-bit.
    """
    y = np.asarray(y, dtype=F32)
    B, n = y.shape
    m = H.shape[0]
    alpha = np.broadcast_to(np.asarray(alpha, dtype=F32), (T,))
    w_in = F32(w_in)
    w_out = F32(w_out)
    chk_vars = [np.flatnonzero(H[c]) for c in range(m)]
    var_chks = [np.flatnonzero(H[:, v]) for v in range(n)]
    cv = {(c, v): np.zeros(B, dtype=F32) for c in range(m) for v in chk_vars[c]}
    outs = [y]
    for it in range(T):
        tot = np.zeros((B, n), dtype=F32)
        for v in range(n):
            acc = np.zeros(B, dtype=F32)
            for c in var_chks[v]:
                acc = acc + cv[(c, v)]
            tot[:, v] = acc + y[:, v] * w_in
        new = {}
        for c in range(m):
            vs = chk_vars[c]
            vc = np.stack([tot[:, v] - cv[(c, v)] for v in vs], axis=1)      # [B,deg]
            s = np.sign(vc).astype(F32)
            S = np.prod(s, axis=1, dtype=F32)
            a = np.minimum(np.abs(vc), F32(1e30))
            srt = np.sort(a, axis=1)
            m1, m2 = srt[:, 0], srt[:, 1]
            for e, v in enumerate(vs):
                mag = np.where(a[:, e] > m1, m1, m2)
                new[(c, v)] = (alpha[it] * mag * (S * s[:, e])).astype(F32)
        cv = new
        out = np.zeros((B, n), dtype=F32)
        for v in range(n):
            acc = np.zeros(B, dtype=F32)
            for c in var_chks[v]:
                acc = acc + cv[(c, v)]
            out[:, v] = acc + w_out * y[:, v]
        outs.append(out)
    return outs


def hard_decision(soft):
    """bit = 0 iff soft > 0; a value of exactly 0 decodes to 1.  (ms_test.py:39)"""
    return np.where(np.asarray(soft) > 0, 0, 1).astype(np.int64)


def evaluate(soft_final, labels, H):
    """``Decoding_model.get_eval`` (ms_test.py:36-54).

    Returns (FER, BER, undetected_count, failed_index) where failed_index are the frames
    with a non-zero syndrome, ascending (what ``tf.where(syndrome!=0)`` yields, :51).
    """
    hard = hard_decision(soft_final)
    err_bits = (hard != labels).sum(axis=1)
    syndrome = (hard.dot(H.T) % 2).sum(axis=1)
    ok = err_bits == 0
    undetected = int(np.count_nonzero((syndrome == 0) & ~ok))
    B, n = labels.shape
    fer = 1.0 - np.count_nonzero(ok) / B
    ber = err_bits.sum() / (B * n)
    return fer, ber, undetected, np.flatnonzero(syndrome != 0)


def collect_failed(outs, labels, failed_index):
    """``collect_failed_output_selective`` (ms_test.py:55-64): T+1 rows per failed frame,
    row 0 = channel values, row j = soft output after iteration j; label repeated."""
    rows, labs = [], []
    for i in failed_index:
        for o in outs:
            rows.append(o[i])
            labs.append(labels[i])
    return rows, labs


# --------------------------------------------------------------------------------------
# OSD front end:  PB_OSD/pb_testing.py:268-320  (== FS_OSD/fs_testing.py:270-322)
# --------------------------------------------------------------------------------------


def reliability_order(y):
    """pi_1 = argsort(|y|) descending (pb_testing.py:309-310).  ``tf.argsort`` is not
    stable, so the reference leaves ties unspecified; the build defines them as
    "lower original index first" (stable)."""
    return np.argsort(-np.abs(np.asarray(y, dtype=F32)), kind="stable")


def identify_mrb(G_ordered, k):
    """``identify_mrb`` (pb_testing.py:268-304) on an already column-permuted G.

    Returns (G' = [I | P'] int64, pi_2 int64[n], swaps).
    """
    n = G_ordered.shape[1]
    R, swaps = gf2_eliminate(G_ordered)                       # :272-274
    idx = np.arange(n)
    for a, b in swaps:                                        # :276-281
        idx[a], idx[b] = idx[b], idx[a]
    mrb, lrb = idx[:k], idx[k:]                               # :283, :291
    sm = np.argsort(mrb, kind="stable")                       # :284  (entries are distinct)
    sl = np.argsort(lrb, kind="stable")                       # :292
    P = R[:, k:][:, sl]                                       # :294
    Pp = P[sm, :]                                             # :287-298: Pi^T . P  <=> row r <- row sm[r]
    Gp = np.concatenate([np.eye(k, dtype=np.int64), Pp], axis=1)   # :300
    pi2 = np.concatenate([mrb[sm], lrb[sl]])                  # :302
    return Gp, pi2, swaps


def swapped_info(y, label, G):
    """``swapped_info`` (pb_testing.py:306-320).  Returns (y', label', G', perm, swaps) with
    ``perm[p]`` = original bit index that sits at primed position p (= pi_1[pi_2[p]])."""
    y = np.asarray(y, dtype=F32)
    k = G.shape[0]
    pi1 = reliability_order(y)
    Gp, pi2, swaps = identify_mrb(G[:, pi1], k)
    perm = pi1[pi2]
    return y[perm], np.asarray(label)[perm], Gp, perm, swaps


# --------------------------------------------------------------------------------------
# test error patterns + conventional order-p OSD:  FS_OSD/convention_osd.py:13-76
# --------------------------------------------------------------------------------------


def tep_table(k, order):
    """``generate_teps`` (convention_osd.py:13-38): for w = 0..order, every weight-w support
    in lexicographic ``itertools.combinations`` order, *stably* re-sorted by descending sum
    of the support indices (:19-24), concatenated.  Returns a list of index tuples."""
    table = []
    for w in range(order + 1):
        combos = list(itertools.combinations(range(k), w))
        combos.sort(key=lambda s: -sum(s))
        table.extend(combos)
    return table


def tep_boundaries(k, order):
    """``query_boundary`` (convention_osd.py:39-47): cumulative C(k,w)."""
    out, acc = [], 0
    for w in range(order + 1):
        acc += math.comb(k, w)
        out.append(acc)
    return out


def tep_matrix(k, order):
    t = tep_table(k, order)
    E = np.zeros((len(t), k), dtype=np.int64)
    for r, s in enumerate(t):
        E[r, list(s)] = 1
    return E


def weighted_distance(disc_bits, w):
    """Canonical float32 evaluation order of  sum_p disc[p] * |y'[p]|.

    The reference evaluates this with ``tf.reduce_sum`` over 128 terms
    (convention_osd.py:60-61, pb_testing.py:106/137, fs_testing.py:62) whose internal order
    is not specified.  The build fixes it as:   M   = flipped MRB weights, ascending
    position, sequential;   L_b = parity byte b (positions k+8b .. k+8b+7), ascending,
    sequential from 0;   cost = (((M + L_0) + L_1) + ...) + L_last.   The HIP kernels and
    the C oracle follow exactly this order, so metrics are bit-identical; ``exact_distance``
    gives the order-free float64 value used to show the choice is immaterial.
    """
    disc_bits = np.asarray(disc_bits)
    w = np.asarray(w, dtype=F32)
    n = w.shape[0]
    k = n // 2 if n % 2 == 0 else None
    return _weighted_distance_k(disc_bits, w, k)


def _weighted_distance_k(disc_bits, w, k):
    n = w.shape[0]
    acc = F32(0)
    for p in range(k):
        if disc_bits[p]:
            acc = F32(acc + w[p])
    p = k
    while p < n:
        part = F32(0)
        for q in range(p, min(p + 8, n)):
            if disc_bits[q]:
                part = F32(part + w[q])
        acc = F32(acc + part)
        p += 8
    return acc


def exact_distance(disc_bits, w):
    return float(np.dot(np.asarray(disc_bits, dtype=np.float64), np.asarray(w, dtype=np.float64)))


def convention_osd(yp, labelp, Gp, order, teps=None):
    """``convention_osd_main`` (convention_osd.py:49-76), returning more than the
    reference does: dict(correct, teps_size, phase, best_index, metric, codeword).

    hard' = (y'>0 ? 0 : 1) (:54); candidates = ((TEP + hard'[:k]) % 2) . G' % 2 (:58-59);
    cost = canonical weighted distance (:60-61); first minimum (:63); ``phase`` is the
    order class of the winner if it equals the label, else -1 (:66-74).
    """
    yp = np.asarray(yp, dtype=F32)
    k, n = Gp.shape
    if teps is None:
        teps = tep_matrix(k, order)
    hard = np.where(yp > 0, 0, 1).astype(np.int64)
    cand = ((teps + hard[None, :k]) % 2).dot(Gp) % 2
    disc = (cand + hard[None, :]) % 2
    w = np.abs(yp)
    cost = np.array([_weighted_distance_k(d, w, k) for d in disc], dtype=F32)
    best = int(np.argmin(cost))
    correct = bool(np.array_equal(cand[best], np.asarray(labelp)))
    phase = -1
    if correct:
        for i, b in enumerate(tep_boundaries(k, order)):
            if best < b:
                phase = i
                break
    return dict(correct=correct, teps_size=int(teps.shape[0]), phase=phase, best_index=best,
                metric=cost[best], codeword=cand[best], costs=cost,
                exact_best=int(np.argmin(disc.astype(np.float64).dot(w.astype(np.float64)))))


# --------------------------------------------------------------------------------------
# FS-OSD:  FS_OSD/fs_testing.py:22-64, 129-161
# --------------------------------------------------------------------------------------


def fs_tep_lists(k, order):
    """``generate_sequential_teps`` (fs_testing.py:32-49): for w = 1..order every combination of
    range(k) in lexicographic order with the indicator vector reversed (:45), i.e. support
    {k-1-p}.  Returns a list (per weight) of lists of ascending support tuples."""
    out = []
    for w in range(1, order + 1):
        out.append([tuple(sorted(k - 1 - p for p in c)) for c in itertools.combinations(range(k), w)])
    return out


def fs_osd_frame(yp, labelp, Gp, order, beta=0.1, tau_e=6.5, tau_psc=30.0):
    """One frame of ``fs_osd`` (fs_testing.py:129-161) in the primed domain.

    Returns dict(codeword_ref, metric_ref = what the reference keeps in ``optimal_codeword`` /
    ``w_dmin`` (:148-152), codeword_hit / metric_hit = the candidate that triggered the tau_e stop
    (appended to ``optimal_list`` but never copied to ``optimal_codeword``, :144-146 -- the quirk
    SURVEY A.4 describes), num_teps (:141), fail_ref = the reference's failure test (:162)).
    """
    yp = np.asarray(yp, dtype=F32)
    k, n = Gp.shape
    w = np.abs(yp)
    hard = np.where(yp > 0, 0, 1).astype(np.int64)

    def one_tep(support):                                   # one_tep_compare :51-64
        mrb = hard[:k].copy()
        for p in support:
            mrb[p] ^= 1
        cw = mrb.dot(Gp) % 2
        disc = (cw + hard) % 2
        return int(disc.sum()), cw, _weighted_distance_k(disc, w, k)

    # acquire_pnc_boundary :22-30 -- float32 running sums of the (i+1) least reliable MRB values
    bounds = []
    for i in range(order):
        acc = F32(0)
        for t in range(k - (i + 1), k):
            acc = F32(acc + w[t])
        bounds.append(acc)
    beta_term = F32(beta * (n - k))                         # :138, python float -> f32 in the TF add
    hd, best_cw, w_dmin = one_tep(())                       # all-zero TEP :131
    num_teps = 1
    hit_cw, hit_metric = None, None
    if not hd < tau_e:                                      # :133-135
        for j in range(order):
            if F32(bounds[j] + beta_term) < w_dmin:         # :139
                stop = False
                for support in fs_tep_lists(k, j + 1)[j]:
                    num_teps += 1                           # :141
                    hd, cw, wd = one_tep(support)
                    if hd < tau_e:                          # :143-146
                        hit_cw, hit_metric, stop = cw, wd, True
                        break
                    if hd < tau_psc and wd < w_dmin:        # :147-152
                        w_dmin, best_cw = wd, cw
                if stop:
                    break
            else:                                           # :155-158
                break
    fail_ref = None if labelp is None else bool(np.any(best_cw != np.asarray(labelp)))
    return dict(codeword_ref=best_cw, metric_ref=w_dmin, codeword_hit=hit_cw, metric_hit=hit_metric,
                num_teps=num_teps, fail_ref=fail_ref)


# --------------------------------------------------------------------------------------
# PB-OSD:  PB_OSD/pb_testing.py:35-41, 100-149, 366-500
# --------------------------------------------------------------------------------------


def _sigmoid32(x):
    return (F32(1) / (F32(1) + np.exp(-np.asarray(x, dtype=F32)))).astype(F32)


def pb_osd_frame(yp, labelp, Gp, order, snr_db):
    """One frame of ``pb_osd`` (pb_testing.py:100-149) in the primed domain, written with NumPy's
    exp and SciPy's binom.cdf (what the reference calls, :458,:478) -- i.e. NOT with the
    deterministic float routines of the C oracle; the two agree except for decisions that sit
    within float rounding of a threshold.  Returns dict(codeword, metric, num_teps, best_index,
    comparisons, stop, fail)."""
    import scipy.stats as stats

    yp = np.asarray(yp, dtype=F32)
    k, n = Gp.shape
    w = np.abs(yp)
    hard = np.where(yp > 0, 0, 1).astype(np.int64)
    c4 = F32(-4.0 * (1.0 / 10 ** (snr_db / 10)))                       # :50-52, used as -4*noise_variance
    q = _sigmoid32(c4 * w)
    p1 = F32(np.add.reduce(q[k:], dtype=F32) / F32(n - k))              # mean_lrb_prob :406-411
    pt = F32(np.add.reduce(q[:k], dtype=F32) / F32(k))                  # mean_mrb_prob :463-468
    lrb_mean = F32(np.add.reduce(w[k:], dtype=F32) / F32(n - k))        # beta_acquire :401
    spl = F32(1)
    for i in range(k):                                                  # com_mrb_prob :35-41
        spl = F32(spl * (F32(1) - q[i]))
    niu = float(stats.binom.cdf(order, k, float(pt)))                   # :478, :489
    nmax = sum(math.comb(k, i) for i in range(order + 1))
    p_t_suc, p_t_pro = 0.99 * niu, 0.002 * math.sqrt((1 - niu) / nmax)  # :490-498

    def encode(support):
        mrb = hard[:k].copy()
        for p in support:
            mrb[p] ^= 1
        cw = mrb.dot(Gp) % 2
        disc = (cw + hard) % 2
        return cw, disc, _weighted_distance_k(disc, w, k)

    def rsum(support):
        acc = F32(0)
        for p in sorted(support):
            acc = F32(acc + w[p])
        return acc

    best_cw, _, w_dmin = encode(())                                     # :101-106
    frontier = [(k - 1,)]                                               # :109-110
    num_teps, best_index, comparisons, stop = nmax, 0, 0, 0
    for j in range(nmax - 1):                                           # :120
        sums = [rsum(s) for s in frontier]
        mi = int(np.argmin(np.array(sums, dtype=F32)))                  # :370 first minimum
        comparisons += 1 if len(frontier) == 1 else 2                   # :371-374
        sel = frontier.pop(mi)
        if sel[-1] < k - 1 and len(sel) < order:                        # :381-384
            frontier.append(sel + (k - 1,))
        if len(sel) > 1:                                                # :386-391
            if sel[-1] - sel[-2] > 1:
                frontier.append(sel[:-1] + (sel[-1] - 1,))
        elif sel[-1] - 1 > -1:                                          # :392-396
            frontier.append((sel[-1] - 1,))
        rs = sums[mi]
        w1 = F32(np.exp(F32(c4 * rs)) * spl)                            # tep_prob :441-445
        w2 = F32(F32(1) - w1)
        bt = np.floor(F32(F32(w_dmin - rs) / lrb_mean))                 # beta_acquire :399-405
        beta = 0 if not bt > 0 else (64 if bt > 64 else int(bt))
        bs = F32(0)
        bs = F32(bs + w1 * F32(stats.binom.cdf(beta, n - k, float(p1))))
        bs = F32(bs + w2 * F32(stats.binom.cdf(beta, n - k, 0.5)))
        if float(bs) < p_t_pro:                                         # :129-132
            stop, num_teps = 1, j + 1
            break
        cw, disc, wd = encode(sel)
        if wd < w_dmin:                                                 # :138-149
            best_cw, w_dmin, best_index = cw, wd, j + 1
            ratio = F32(F32(F32(1) - w1) / w1)
            prod = F32(1)
            for i in range(k, n):
                prod = F32(prod * (F32(2) * q[i] if disc[i] else F32(2) * (F32(1) - q[i])))
            p_suc = F32(F32(1) / F32(F32(1) + F32(ratio / prod)))
            if p_suc > F32(p_t_suc):                                    # (:145: TF compares the f32 tensor with the double cast TO f32)
                stop, num_teps = 2, j + 1
                break
    fail = None if labelp is None else bool(np.any(best_cw != np.asarray(labelp)))
    return dict(codeword=best_cw, metric=w_dmin, num_teps=num_teps, best_index=best_index,
                comparisons=comparisons, stop=stop, fail=fail)


# --------------------------------------------------------------------------------------
# DL-OSD stage, H-form primitives:
#   DL_OSD_Testing_serial/ordered_statistics_decoding.py:25-98,153-257, globalmap.py:57-76,
#   nn_testing.py:65-82,120-148
# --------------------------------------------------------------------------------------


def segment_boundaries(k, num_seg):
    """``secure_segment_threshold`` (DL_OSD_Testing_serial/globalmap.py:57-76): a head segment of
    one position, then num_seg-1 segments growing like 1:2:...; -> (sizes, boundaries)."""
    alloc = k - 1
    basic = list(range(1, num_seg))
    sizes = [int(alloc / sum(basic) * b) for b in basic]
    sizes[-1] += alloc - sum(sizes)
    sizes = np.insert(np.array(sizes, dtype=np.int64), 0, 1)
    return sizes, np.insert(np.cumsum(sizes), 0, 0)


def error_pattern_gen(direction, range_list, k):
    """``osd.error_pattern_gen`` (ordered_statistics_decoding.py:81-98): all error patterns with
    ``direction[i]`` flips inside segment ``range_list[i]``; itertools.product over the segments
    of itertools.combinations inside each.  -> int array [N, k]."""
    per_seg = [list(itertools.combinations(range_list[i], v)) if v else [()] for i, v in enumerate(direction)]
    joined = list(itertools.product(*per_seg))
    E = np.zeros((len(joined), k), dtype=np.int64)
    for r, seq in enumerate(joined):
        E[r, list(itertools.chain.from_iterable(seq))] = 1
    return E


def convention_path(order_sum):
    """``query_convention_path`` (nn_testing.py:65-82): 3-segment order patterns with total <= i
    for i = 0..order_sum, first occurrence kept (UniqueV2 keeps first-seen order)."""
    path, seen = [], set()
    for i in range(order_sum + 1):
        for j1 in range(order_sum + 1):
            for j2 in range(order_sum + 1):
                for j3 in range(order_sum + 1):
                    if j1 + j2 + j3 <= i and (j1, j2, j3) not in seen:
                        seen.add((j1, j2, j3))
                        path.append([j1, j2, j3])
    return path


def hosd_reorder(order_llr):
    """``mag_input_gen`` (:25-28): argsort(|x|) ASCENDING; ties -> lower index first (defined here,
    tf.argsort leaves it open)."""
    return np.argsort(np.abs(np.asarray(order_llr, dtype=F32)), kind="stable")


def hosd_identify_mrb(H_ordered, k):
    """``osd.identify_mrb`` (:43-80) for one frame, on H with columns already in ascending
    reliability order.  -> (updated_index_order[n], updated_M[m,k], swaps).  The LRB part keeps
    the order the elimination left it in (:69); only the MRB part is sorted (:66-70)."""
    m, n = H_ordered.shape
    R, swaps = gf2_eliminate(H_ordered)                        # :55 (same rule as full_gf2elim :222-257)
    if R.shape[0] != m:
        raise ValueError("rank-deficient H")
    idx = np.arange(n)
    for a, b in swaps:                                         # :58-62
        idx[a], idx[b] = idx[b], idx[a]
    mrb = idx[-k:]                                             # :64
    sw = np.argsort(mrb, kind="stable")                        # :66
    uidx = np.concatenate([idx[: n - k], mrb[sw]])             # :67-68
    M = R[:, -k:][:, sw]                                       # :69
    return uidx, M, swaps


def hosd_cost(disc, w):
    """Canonical float32 order of sum_p disc[p]*w[p] over the 128 updated positions (LRB first):
    bytes of 8 positions, each summed ascending from 0, then the 16 byte sums added ascending
    starting from byte 0.  ``disc`` may be [N,128] (vectorised over candidates)."""
    disc = np.atleast_2d(np.asarray(disc)).astype(F32)
    w = np.asarray(w, dtype=F32)
    acc = None
    for b in range(disc.shape[1] // 8):
        part = np.zeros(disc.shape[0], dtype=F32)
        for t in range(8):
            part = (part + disc[:, 8 * b + t] * w[8 * b + t]).astype(F32)
        acc = part if acc is None else (acc + part).astype(F32)
    return acc


def hosd_frame(order_llr, metric_llr, label, H, blocks):
    """One frame of the block evaluation inside ``osd.sliding_osd`` (:164-186) with every block of
    ``blocks`` (list of [N_b,k] pattern matrices) evaluated by ``acquire_min`` (:153-162).

    order_llr sorts the positions and supplies the starting MRB hard decisions (:174,183-184);
    metric_llr (trajectory row 0) supplies the hard decisions and weights of the metric
    (:175,180-182).  -> dict(lri, uidx, perm, M, swaps, block_min, block_arg, truth, best_index,
    metric, codeword[n] in ORIGINAL bit order)."""
    order_llr = np.asarray(order_llr, dtype=F32)
    metric_llr = np.asarray(metric_llr, dtype=F32)
    m, n = H.shape
    k = n - m
    lri = hosd_reorder(order_llr)                              # :34
    uidx, M, swaps = hosd_identify_mrb(np.asarray(H)[:, lri], k)
    perm = lri[uidx]
    o_in, o_orig = order_llr[perm], metric_llr[perm]           # :172-173
    hard_orig = np.where(o_orig > 0, 0, 1).astype(np.int64)    # :180
    mag = np.abs(o_orig)                                       # :182
    initial_mrb = np.where(o_in > 0, 0, 1).astype(np.int64)[-k:]   # :186-187
    mins, args, off = [], [], 0
    best = (None, -1, None)
    for E in blocks:
        mrb = (np.asarray(E, dtype=np.int64) + initial_mrb[None, :]) % 2      # :154
        lrb = mrb.dot(M.T) % 2                                                 # :155
        cand = np.concatenate([lrb, mrb], axis=1)                              # :156
        cost = hosd_cost((cand + hard_orig[None, :]) % 2, mag)                 # :159-160
        a = int(np.argmin(cost)) if len(cost) else -1
        mins.append(cost[a] if a >= 0 else F32(np.inf))
        args.append(off + a if a >= 0 else -1)
        if a >= 0 and (best[0] is None or cost[a] < best[0]):
            best = (cost[a], off + a, cand[a])
        off += len(cost)
    truth = None
    if label is not None:
        lab = np.asarray(label, dtype=np.int64)[perm]          # :176
        truth = hosd_cost((lab + hard_orig) % 2, mag)[0]       # :181-183
    cw = np.zeros(n, dtype=np.int64)
    if best[2] is not None:
        cw[perm] = best[2]
    return dict(lri=lri, uidx=uidx, perm=perm, M=M, swaps=swaps, block_min=np.array(mins, dtype=F32),
                block_arg=np.array(args, dtype=np.int64), truth=truth, best_index=best[1], metric=best[0],
                codeword=cw)


def sliding_window_decide(block_min, truth, fcn, win, soft_margin, acc_block_size):
    """The window loop of ``osd.sliding_osd`` (:187-218) replayed on precomputed block minima.
    ``fcn(x[1, win+1]) -> [p0, p1]``.  -> (success, windows, complexity, global_min)."""
    window = list(block_min[:win])                             # :188-191
    global_min = min(window)
    deep_limit = win
    for kk in range(len(block_min) - win + 1):                 # :193
        deep_limit = kk + win
        if kk != 0:
            ms = block_min[win + kk - 1]                       # :198-199
            window.append(ms)
            window = window[-win:]
            if ms > global_min:                                # :203-204
                continue
        sw = np.sort(np.asarray(window, dtype=F32))            # sliding_window_ops :140-151
        prob = np.asarray(fcn(np.append(sw, F32(kk)).reshape(1, -1))).reshape(-1)
        global_min = min(global_min, min(window))
        if prob[1] > soft_margin:
            break
    windows = deep_limit - win + 1                             # :210
    return bool(global_min == truth), windows, int(acc_block_size[deep_limit]), global_min
