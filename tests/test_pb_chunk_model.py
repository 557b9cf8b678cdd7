"""The sorted-chunk formulation of PB-OSD (tests/pb_chunk_model.py = the algorithm of csrc/ldpc_osd_pb.hip in NumPy)
against the literal frontier-list restatement of the C oracle (pb_testing.py:100-149, :366-397): TEP counts, stop
reasons, frontier comparisons, both success counters, winner index and metric must agree on every frame the
chunk model accepts; frames it declines (massive ties) are the ones the kernel hands to the list replay."""
import numpy as np
import pytest

from oracle import c_oracle, np_oracle
from tests import pb_chunk_model as M


def _failed(np_code, snr, frames, seed, quant=None):
    rng = np.random.default_rng(seed)
    y, cw = np_oracle.make_frames(np_code.G, snr, frames, rng)
    if quant:
        y = (np.round(y * quant) / quant).astype(np.float32)
    soft = c_oracle.nms(np_code.H, y, 10, 0.669435)
    _, fail, _ = c_oracle.evaluate(np_code.H, soft, cw)
    idx = np.flatnonzero(fail)
    return y[idx], cw[idx]


@pytest.mark.parametrize("snr,order,quant,kw", [
    (2.5, 2, None, {}), (2.5, 3, None, dict(cap=2048, nbins=1024, schedule=(0,))), (1.0, 3, None, dict(cap=2048, nbins=1024, schedule=(0,))),
    (2.5, 1, None, {}), (2.5, 3, 64.0, dict(cap=2048, nbins=1024, schedule=(0,))), (2.5, 2, 8.0, {}),
])
def test_chunk_model_equals_list_replay(np_code, snr, order, quant, kw):
    y, cw = _failed(np_code, snr, 400 if snr > 2 else 80, seed=int(snr * 10) + order, quant=quant)
    y, cw = y[:40], cw[:40]
    ref = c_oracle.pb_osd(np_code.G, y, cw, order, snr)
    tabs = M.pb_table(order)
    accepted = 0
    for j in range(y.shape[0]):
        o = M.pb_chunk_frame(M.frame_inputs(np_code.G, y[j]), order, snr, tables=tabs, **kw)
        if o == M.DEGENERATE:
            continue
        accepted += 1
        got = (o["ntep"], o["stop"], o["cmp"], o["suc1"], o["suc2"], o["bestidx"], o["best"])
        want = tuple(ref[k][j] for k in ("num_teps", "stop", "comparisons", "suc1", "suc2", "best_index", "metric"))
        assert got == want, (j, got, want)
    assert accepted >= (y.shape[0] // 2 if quant is None else 1)


def test_visit_order_comparator_on_ties():
    """Equal sums: list order = order of the parents' pops, the extended child before the adjacent one."""
    w = np.ones(128, dtype=np.float32)          # every weight-w TEP has the same sum
    assert M.visit_less(w, (63,), (62,)) and not M.visit_less(w, (62,), (63,))
    assert M.visit_less(w, (61,), (60,))
    w2 = w.copy(); w2[63] = 0.0                   # {p} and {p, 63} tie: parent before child
    assert M.visit_less(w2, (10,), (10, 63)) and not M.visit_less(w2, (10, 63), (10,))
    assert M.visit_less(w2, (62, 63), (61,))     # both children of {62}, equal sums: the extended one first


# ---------------------------------------------------------------------------------------------------------------------
# The direct enumeration of a sum range (csrc/ldpc_osd_pb.hip, PbItems): item layout, table-id formulas, the binary
# search and the empty-item shortcut, restated in NumPy float32 and checked against a brute-force filter of the kernel's
# TEP table.
# ---------------------------------------------------------------------------------------------------------------------
def _device_table():
    tab = [(p, 0, 0, 1) for p in range(63, -1, -1)]
    tab += [(p0, p1, 0, 2) for p0 in range(62, -1, -1) for p1 in range(p0 + 1, 64)]
    tab += [(p0, p1, p2, 3) for p0 in range(61, -1, -1) for p1 in range(p0 + 1, 63) for p2 in range(p1 + 1, 64)]
    return np.array(tab, dtype=np.int64)


def _first_le(w, sb, base, T):
    lo, hi = base + 1, 64
    for _ in range(7):
        mid = (lo + hi) >> 1
        act, ok = lo < hi, np.float32(sb + w[min(mid, 63)]) <= T
        if act and ok:
            hi = mid
        elif act:
            lo = mid + 1
    return lo


def _enumerate(tab, w, lo, hi, order):
    nitems = 2080 if order > 2 else (64 if order > 1 else 1)
    out = []
    for it in range(nitems):
        if it == 0:
            sb, base, idb = np.float32(0), -1, None
        elif it < 64:
            i = it - 1
            m = 63 - i
            sb, base, idb = w[i], i, 64 + m * (m - 1) // 2 - (i + 1)
        else:
            i, j = int(tab[it][0]), int(tab[it][1])
            m, r = 63 - i, 64 - j
            sb, base = np.float32(w[i] + w[j]), j
            idb = 2080 + m * (m - 1) * (m - 2) // 6 + m * (m - 1) // 2 - r * (r - 1) // 2 - (j + 1)
        e = 64 if (lo < 0 or base >= 63) else _first_le(w, sb, base, lo)
        a = e
        if base + 1 < e and np.float32(sb + w[63]) <= hi:
            a = min(_first_le(w, sb, base, hi), e)
        for mm in range(a, e):
            out.append((63 - mm if base < 0 else idb + mm, np.float32(sb + w[mm])))
    return out


@pytest.mark.parametrize("order", [1, 2, 3])
def test_sum_range_enumeration(order):
    tab = _device_table()
    nall = {1: 64, 2: 2080, 3: 43744}[order]
    rng = np.random.default_rng(5 + order)
    for case in range(6):
        w = np.sort(np.abs(rng.normal(1.0, 0.6, 64)).astype(np.float32))[::-1].copy()
        if case == 1:
            w = (np.round(w * 8) / 8).astype(np.float32)      # many equal values
        if case == 2:
            w[40:] = 0.0
        s = w[tab[:nall, 0]].copy()
        s[tab[:nall, 3] > 1] = (s + w[tab[:nall, 1]])[tab[:nall, 3] > 1]
        s[tab[:nall, 3] > 2] = (s + w[tab[:nall, 2]])[tab[:nall, 3] > 2]
        bounds = [np.float32(-1.0), w[0], np.float32(w[0] * 1.4), np.float32(w[0] * 2.2), np.float32(np.inf)]
        seen = np.zeros(nall, dtype=bool)
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            got = _enumerate(tab, w, lo, hi, order)
            ids = np.array([g[0] for g in got], dtype=np.int64)
            want = np.flatnonzero((s > lo) & (s <= hi))
            assert np.array_equal(np.sort(ids), want), (order, case, lo, hi)
            assert all(np.float32(v) == s[i] for i, v in got)        # the sums are the table's sums, bit for bit
            assert not seen[ids].any()
            seen[ids] = True
        assert seen.all()


def _items():
    """The 2017 items of csrc/ldpc_osd_pb.hip (pbw_item): item (row q, lane l) -> fixed positions and the base of its members
    (members are the last positions m in (base, 63])."""
    out = {}
    for q in range(32):
        for l in range(64):
            if q < 31:
                if l == 63:
                    continue                                   # lane 63 owns no triple item
                first = l < 62 - q                             # rows by the distance of the two fixed positions (round 4)
                i, j = (l, l + 1 + q) if first else (l - 62 + q, l)
                out[(q, l)] = ((i, j), j)
            else:
                out[(q, l)] = ((l,), l) if l <= 62 else ((), -1)   # the pairs of i = l; lane 63: the singles
    return out


def test_items_partition_the_tep_table():
    """Every TEP of weight 1..3 over 64 positions is a member of exactly one item (the walk of the chunk kernel and the
    bisection counts of the workgroup kernel enumerate items, not TEPs)."""
    seen = set()
    for (q, l), (fixed, base) in _items().items():
        for m in range(base + 1, 64):
            t = tuple(sorted(fixed + (m,)))
            assert len(set(t)) == len(t) and t not in seen, (q, l, t)
            seen.add(t)
    assert len(seen) == 64 + 2016 + 41664


def test_workgroup_kernel_dealing_is_a_bijection():
    """coop_item (csrc/ldpc_osd_pb.hip): item j of lane `lane` in wavefront v is row q = 16 j + lane / 4, lane
    l = (13 (v - 3 q) mod 16) + 16 (lane mod 4), i.e. the items with (5 l + 3 q) mod 16 = v: every (q, l) exactly once."""
    seen = set()
    for v in range(16):
        for j in range(2):
            for lane in range(64):
                q = 16 * j + (lane >> 2)
                l = ((13 * (v - 3 * q)) & 15) + 16 * (lane & 3)
                assert (5 * l + 3 * q) % 16 == v
                assert (q, l) not in seen
                seen.add((q, l))
    assert len(seen) == 32 * 64
    # the eight-wavefront form of the template (two frames per CU; measured, not the product setting): row q = 8 j + lane / 8,
    # lane l = (5 (v - 3 q) mod 8) + 8 (lane mod 8), the items with (5 l + 3 q) mod 8 = v
    seen = set()
    for v in range(8):
        for j in range(4):
            for lane in range(64):
                q = 8 * j + (lane >> 3)
                l = ((5 * (v - 3 * q)) & 7) + 8 * (lane & 7)
                assert (5 * l + 3 * q) % 8 == v
                assert (q, l) not in seen
                seen.add((q, l))
    assert len(seen) == 32 * 64


def test_workgroup_kernel_dealing_balances_a_chunk(np_code):
    """What the dealing is for: over the depths of a search, the sixteen wavefronts generate about the same number of a
    chunk's keys (whole rows per wavefront: the fullest holds 1.4-1.9 times the mean and the others wait for it)."""
    y, _ = _failed(np_code, 2.5, 200, seed=3)
    items = _items()
    worst = []
    for fr in y[:6]:
        w = np.sort(np.abs(fr))[::-1][:64].astype(np.float32)          # stand-in for the MRB reliabilities, descending
        sums, owner = [], []
        for (q, l), (fixed, base) in items.items():
            sb = np.float32(sum(w[p] for p in fixed)) if fixed else np.float32(0)
            for m in range(base + 1, 64):
                sums.append(sb + w[m]); owner.append((5 * l + 3 * q) % 16)
        order = np.argsort(np.array(sums), kind="stable")
        owner = np.array(owner)[order]
        for lo in range(1024, 40000, 2700):
            c = np.bincount(owner[lo:lo + 2700], minlength=16)
            worst.append(c.max() / (2700 / 16))
    assert np.mean(worst) < 1.2 and max(worst) < 1.5


def test_bisection_count_equals_enumeration():
    """coop_count (csrc/ldpc_osd_pb.hip): inside an item the float32 sums sb + w[m] fall as m rises (w is sorted descending,
    float addition is monotone), so the members <= T beyond the cursor a are the positions [first, a) and seven bisection
    steps find `first` for any base in [-1, 63] and cursor in [base + 1, 64] -- also with equal reliabilities."""
    rng = np.random.default_rng(11)
    for trial in range(300):
        w = np.sort(np.abs(rng.standard_normal(64)).astype(np.float32))[::-1].copy()
        if trial % 3 == 0:
            w = (np.round(w * 8) / 8).astype(np.float32)              # ties
        sb = np.float32(rng.uniform(0, 3))
        base = int(rng.integers(-1, 64))
        a = int(rng.integers(base + 1, 65))
        T = np.float32(sb + rng.uniform(0, 3))
        lo, hi = base + 1, a
        for _ in range(7):
            act = lo < hi
            mid = (lo + hi) >> 1
            f = np.float32(sb + w[mid & 63]) <= T
            hi = mid if act and f else hi
            lo = mid + 1 if act and not f else lo
        want = [m for m in range(base + 1, a) if np.float32(sb + w[m]) <= T]
        assert lo >= hi, (base, a)
        assert want == list(range(hi, a)), (trial, base, a, hi, want[:3])
