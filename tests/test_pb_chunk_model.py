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
