"""-m gpu: PB-OSD kernel (pb_testing.py:100-149) through the C ABI against the C oracle.
Both sides use the same deterministic float routines, so TEP counts, stop reasons, winners and
metrics must agree exactly."""
import numpy as np
import pytest
import torch

from oracle import c_oracle, np_oracle
from tests.gpu_util import pack_np, to_dev, words_np

pytestmark = pytest.mark.gpu
ALPHA0 = 0.669435


@pytest.fixture(scope="module")
def dec():
    from short_ldpc_decoding_osd_amd import Code
    from short_ldpc_decoding_osd_amd.runtime import Decoder
    return Decoder(Code())


def _failures(dec, snr, frames, seed):
    rng = np.random.default_rng(seed)
    y, cw = np_oracle.make_frames(dec.code.G, snr, frames, rng)
    soft = c_oracle.nms(dec.code.H, y, 10, ALPHA0)
    _, fail, _ = c_oracle.evaluate(dec.code.H, soft, cw)
    idx = np.flatnonzero(fail)
    return y[idx], cw[idx]


def _check(dec, y, cw, order, snr, path, ref=None):
    from short_ldpc_decoding_osd_amd import _lib
    ref = ref if ref is not None else c_oracle.pb_osd(dec.code.G, y, cw, order, snr)
    aux = torch.zeros((y.shape[0], 4), dtype=torch.int32, device=dec.device)
    p = dec.osd_params(order, _lib.OSD_PB, snr_db=snr, aux=aux, pb_path=path)
    out = dec.osd_decode(to_dev(y, dec), order, params=p)
    torch.cuda.synchronize()
    a = aux.cpu().numpy()
    bad = np.flatnonzero(out["ntep"].cpu().numpy() != ref["num_teps"])
    assert bad.size == 0, (path, bad[:8], out["ntep"].cpu().numpy()[bad[:8]], ref["num_teps"][bad[:8]], a[bad[:8]], ref["stop"][bad[:8]])
    assert np.array_equal(a[:, 3], ref["stop"])
    assert np.array_equal(a[:, 0], ref["comparisons"]) and np.array_equal(a[:, 1], ref["suc1"])
    assert np.array_equal(a[:, 2], ref["suc2"])
    assert np.array_equal(out["best"].cpu().numpy(), ref["best_index"])
    assert np.array_equal(words_np(out["cw"]), pack_np(ref["codeword"]))
    assert np.array_equal(out["metric"].cpu().numpy(), ref["metric"])
    return ref


# the three routes of the PB-OSD launcher: staged (weight-1 head per wavefront, then sorted chunks per workgroup, list replay
# for massive ties), every frame through the workgroup kernel, every frame through the literal list replay
PATHS = [None, "block", "replay"]


@pytest.mark.parametrize("snr,order,frames", [(2.5, 2, 3000), (2.5, 3, 1500), (1.0, 2, 600), (3.5, 3, 6000), (2.5, 1, 800), (2.0, 3, 900), (3.0, 3, 3000)])
def test_pb_matches_oracle(dec, snr, order, frames):
    y, cw = _failures(dec, snr, frames, seed=int(snr * 10) + order)
    y, cw = y[:600], cw[:600]
    ref = None
    for path in PATHS:
        ref = _check(dec, y, cw, order, snr, path, ref)
    # SURVEY 6 scale check: PB-OSD visits ~1e2 TEPs per frame, far below the 2081 / 43745 of the full scan
    assert ref["num_teps"].mean() < 1000


def test_pb_full_scan_and_spill(dec):
    """A frame on which no rule fires runs all N_max - 1 TEPs (frontier spills past the LDS head)."""
    from short_ldpc_decoding_osd_amd import _lib
    y, cw = _failures(dec, 2.5, 1500, seed=8)
    y, cw = y[:64], cw[:64]
    # snr_db = 40 dB makes every bit-error probability tiny: the success rule needs a near-perfect match
    for snr in (-20.0, 40.0):
        ref = None
        for path in PATHS:
            ref = _check(dec, y, cw, 2, snr, path, ref)


def test_pb_long_searches_order3(dec):
    """1.0 dB, order 3: searches of thousands of TEPs -- many chunks per frame (bounds sized by exact counts), the +inf bound, full scans."""
    y, cw = _failures(dec, 1.0, 120, seed=77)
    y, cw = y[:48], cw[:48]
    ref = _check(dec, y, cw, 3, 1.0, None)
    assert ref["num_teps"].max() > 8000
    _check(dec, y, cw, 3, 1.0, "block", ref)
    y2, cw2 = _failures(dec, 2.5, 300, seed=78)
    ref2 = _check(dec, y2[:24], cw2[:24], 3, -5.0, None)       # (almost) no rule fires: all 43 744 TEPs of a frame
    assert (ref2["stop"] == 0).sum() >= 12 and (ref2["stop"] != 0).any()
    _check(dec, y2[:24], cw2[:24], 3, -5.0, "block", ref2)


@pytest.mark.parametrize("quant", [4.0, 16.0, 256.0])
def test_pb_tie_heavy_inputs(dec, quant):
    """Quantised channel values make reliability sums tie exactly: the pop order is then decided by list order
    (slot numbers), which the chunk kernels rebuild through the parents -- or hand the frame to the list replay."""
    rng = np.random.default_rng(int(quant))
    y, cw = np_oracle.make_frames(dec.code.G, 2.5, 1200, rng)
    y = (np.round(y * quant) / quant).astype(np.float32)
    soft = c_oracle.nms(dec.code.H, y, 10, ALPHA0)
    _, fail, _ = c_oracle.evaluate(dec.code.H, soft, cw)
    idx = np.flatnonzero(fail)[:160]
    for order in (2, 3):
        ref = _check(dec, y[idx], cw[idx], order, 2.5, None)
        _check(dec, y[idx], cw[idx], order, 2.5, "block", ref)


@pytest.mark.parametrize("scale", [16.0, 1.0 / 64.0, 1000.0])
def test_pb_other_input_scalings(dec, scale):
    """The chunk bounds are sized relative to the frame's own reliabilities (pb_pick_bound): other scalings of the
    channel values (LLR-scaled inputs) with the SNR parameter moved along take the same route through the kernels."""
    y, cw = _failures(dec, 2.5, 1200, seed=91)
    y, cw = (y[:200] * np.float32(scale)).astype(np.float32), cw[:200]
    snr = 2.5 + 10.0 * np.log10(scale)
    ref = _check(dec, y, cw, 3, snr, None)
    assert ref["num_teps"].max() > 1000
    _check(dec, y, cw, 3, snr, "block", ref)


def test_pb_workgroup_kernel_overflowing_hand_over(dec):
    """The workgroup kernel takes at most 2 x 4096 searches a call (its list has two halves: searches expected to run far, served
    first, and the others): with the hand-over budget forced to 64 TEPs (the context's tuning, ldpc_ctx_set_pb_tuning) more than
    that ask to leave the chunk kernel, so at least one half overflows -- the searches that find room are finished by the
    workgroup kernel from their first chunks on, the others stay where they are.  Same counts, stops, winners and metrics."""
    prev = dec.set_pb_tuning(budget_s=64, budget_m=64, budget=64, budget_l=64, budget_xl=64)
    try:
        assert dec.pb_tuning()["budget_xl"] == 64
        y, cw = _failures(dec, 1.5, 30000, seed=5)
        y, cw = y[:18000], cw[:18000]
        assert y.shape[0] == 18000
        ref = _check(dec, y, cw, 3, 1.5, None)
        assert (ref["num_teps"] > 64).sum() > 2 * 4096 + 200
    finally:
        dec.set_pb_tuning(**prev)
    assert dec.pb_tuning() == prev


def test_pb_tuning_is_validated(dec):
    """Bad values are refused and change nothing (LDPC_PB_LATE_DIV=0 used to reach the kernel); no field = defaults."""
    from short_ldpc_decoding_osd_amd import _lib
    before = dec.pb_tuning()
    for bad in (dict(late_div=0), dict(budget=0), dict(t2=833), dict(t1=8), dict(t3=5000), dict(budget_xl=-1)):
        with pytest.raises(_lib.LdpcError):
            dec.set_pb_tuning(**bad)
        assert dec.pb_tuning() == before
    with pytest.raises(ValueError):
        dec.set_pb_tuning(no_such_field=1)
    dec.set_pb_tuning(t2=256, late_pct=65, late_div=2)
    assert dec.pb_tuning()["t2"] == 256
    dec.set_pb_tuning()
    assert dec.pb_tuning() == before == dict(budget=4096, budget_s=512, budget_m=1024, budget_l=8192, budget_xl=24576, t1=320, t2=600,
                                             t3=3072, late_min=4608, late_maxlen=1 << 30, late_pct=20, late_div=16, handoff_maxlen=1 << 30)


@pytest.mark.parametrize("tuning", [dict(t1=32, t2=32), dict(t1=832, t2=832), dict(t1=64, t2=800, budget_s=100000, budget_m=100000, budget=100000),
                                    dict(t1=500, t2=97, budget_s=300, budget_m=300, budget=300, t3=256),
                                    dict(late_pct=0, late_div=64), dict(late_pct=50, late_div=8, t2=200), dict(late_div=1)])
def test_pb_results_do_not_depend_on_the_tuning(dec, tuning):
    """Chunk targets at both ends of their range (tiny chunks; chunks that overflow the 832-key buffer and the work-list ring and
    are retried; a chunk extended in place), with and without hand-over to the workgroup kernel: counts, stops, winners and
    metrics stay exact (the bounds of a chunk are free parameters of the method).  The last three: the tail rule of the chunk
    kernel (searches leave sooner once their sub-list has started) from the first frame on / from half way / off."""
    prev = dec.set_pb_tuning(**tuning)
    try:
        y, cw = _failures(dec, 1.5, 900, seed=21)
        _check(dec, y[:400], cw[:400], 3, 1.5, None)
        y2, cw2 = _failures(dec, 2.5, 1200, seed=22)
        _check(dec, y2[:200], cw2[:200], 2, 2.5, None)
    finally:
        dec.set_pb_tuning(**prev)


@pytest.mark.parametrize("quant", [1024.0, 16384.0])
def test_pb_workgroup_kernel_ties(dec, quant):
    """Long searches on finely quantised channel values: equal sums deep inside a search, where the workgroup kernel's
    sort-free pass meets a tie against a reference key and wavefront 0 redoes the chunk with the sorted path."""
    prev = dec.set_pb_tuning(budget_s=128)
    try:
        rng = np.random.default_rng(int(quant) + 1)
        y, cw = np_oracle.make_frames(dec.code.G, 1.0, 400, rng)
        y = (np.round(y * quant) / quant).astype(np.float32)
        soft = c_oracle.nms(dec.code.H, y, 10, ALPHA0)
        _, fail, _ = c_oracle.evaluate(dec.code.H, soft, cw)
        idx = np.flatnonzero(fail)[:96]
        ref = _check(dec, y[idx], cw[idx], 3, 1.0, None)
        assert ref["num_teps"].max() > 8000
    finally:
        dec.set_pb_tuning(**prev)


@pytest.mark.parametrize("snr,order,quant", [(2.5, 3, 0.0), (1.0, 3, 0.0), (2.0, 2, 0.0), (1.5, 3, 512.0)])
def test_pb_front_end_inside_the_first_kernel(dec, snr, order, quant):
    """With params.reserved bit 0, ldpc_osd_decode runs the OSD front end inside the PB singles kernel (round 4: no workspace
    between them; a frame of the list replay is then set up from its singles record); by default the front end is a kernel of
    its own, and ldpc_osd_front + ldpc_osd_search is the third way to the same search: all three agree word for word, on an
    index list too.  (quant: quantised channel values -- massive ties, the list replay.)"""
    from short_ldpc_decoding_osd_amd import _lib
    y, cw = _failures(dec, snr, 2500, seed=int(snr * 10) + 40 + order)
    if quant:
        y = (np.round(y * quant) / quant).astype(np.float32)
    y, cw = y[:700], cw[:700]
    yd = to_dev(y, dec)
    outs = []
    for kw in (dict(), dict(pb_front_inside=True)):
        aux = torch.zeros((y.shape[0], 4), dtype=torch.int32, device=dec.device)
        o = dec.osd_decode(yd, order, params=dec.osd_params(order, _lib.OSD_PB, snr_db=snr, aux=aux, **kw))
        outs.append({k: o[k].clone() for k in ("cw", "metric", "best", "ntep")} | {"aux": aux})
    aux = torch.zeros((y.shape[0], 4), dtype=torch.int32, device=dec.device)
    perm, parity, _ = dec.osd_front(yd)
    o = dec.osd_search(yd, perm, parity, dec.osd_params(order, _lib.OSD_PB, snr_db=snr, aux=aux))
    outs.append({k: o[k].clone() for k in ("cw", "metric", "best", "ntep")} | {"aux": aux})
    torch.cuda.synchronize()
    for other in outs[1:]:
        for k in outs[0]:
            assert torch.equal(outs[0][k], other[k]), k
    # an index list over a larger batch: every third frame, in reverse
    idx = torch.arange(y.shape[0] - 1, -1, -3, device=dec.device, dtype=torch.int32).contiguous()
    a = dec.osd_decode(yd, order, index=idx, params=dec.osd_params(order, _lib.OSD_PB, snr_db=snr, pb_front_inside=True))
    torch.cuda.synchronize()
    sel = idx.long()
    for k in ("cw", "metric", "best", "ntep"):
        assert torch.equal(a[k][: idx.numel()], outs[0][k][sel]), k


@pytest.mark.parametrize("quant,picks", [(1024.0, (293,)), (64.0, (344, 371))])
def test_pb_two_records_on_one_sum(dec, quant, picks):
    """Found by tests/tools/pb_long_fuzz.py (round 4): two improvement candidates with EQUAL sums of which the second stops by the
    success rule.  The workgroup kernel orders equal sums by the visit comparator since round 4 and counted "records with a sum
    below the stopping record's" -- one short when the record before it ties with it: it returned the earlier record as the
    winner (TEP count, metric and counters of that one).  The records are in visit order: all of them count."""
    snr, order = 1.5, 3
    y, cw = np_oracle.make_frames(dec.code.G, snr, 823, np.random.default_rng(951034177))
    y = (np.round(y * quant) / quant).astype(np.float32)
    soft = c_oracle.nms(dec.code.H, y, 10, ALPHA0)
    _, fail, _ = c_oracle.evaluate(dec.code.H, soft, cw)
    idx = np.flatnonzero(fail)[list(picks)]
    for tuning in (dict(), dict(budget_s=64, budget_m=64, budget=64, budget_l=64, budget_xl=64)):
        prev = dec.set_pb_tuning(**tuning)
        try:
            ref = None
            for path in PATHS:
                ref = _check(dec, y[idx], cw[idx], order, snr, path, ref)
            assert (ref["suc2"] >= 6).all() and (ref["stop"] == 2).all()
        finally:
            dec.set_pb_tuning(**prev)


@pytest.mark.parametrize("tuning", [dict(budget_s=64, budget_m=64, budget=64), dict(budget_s=700, budget_m=700, budget=700),
                                    dict(late_pct=100000, late_div=64, late_min=0)])
def test_pb_ties_under_schedules(dec, tuning):
    """Equal sums against reference keys inside the workgroup kernel's sort-free pass (ordered by the visit comparator since
    round 4, not handed to one wavefront): quantised channel values, searches handed on after 64 / 700 TEPs and by the tail
    rule from the first chunk on.  (tests/tools/pb_tie_stress.py: the long form, 8640 decodes.)"""
    prev = dec.set_pb_tuning(**tuning)
    try:
        for quant, snr, order in ((256.0, 1.0, 3), (65536.0, 2.0, 3), (64.0, 1.5, 2)):
            rng = np.random.default_rng(int(quant) + 7)
            y, cw = np_oracle.make_frames(dec.code.G, snr, 300, rng)
            y = (np.round(y * quant) / quant).astype(np.float32)
            soft = c_oracle.nms(dec.code.H, y, 10, ALPHA0)
            _, fail, _ = c_oracle.evaluate(dec.code.H, soft, cw)
            idx = np.flatnonzero(fail)[:64]
            _check(dec, y[idx], cw[idx], order, snr, None)
    finally:
        dec.set_pb_tuning(**prev)


@pytest.mark.parametrize("snr,B", [(2.5, 131072), (1.0, 131072), (1.75, 131072)])
def test_pb_full_size_properties(dec, snr, B):
    """BASELINE config 5 sizes: size-independent properties of PB-OSD order 3 on the NMS failures of a full batch.
    The chunk bounds, the hand-over to the workgroup kernel and the order of the keys inside a chunk depend on the batch
    (list lengths) and on timing; the results must not: two runs agree bit for bit, and so does the same set of frames
    decoded in slices of 3000 (other budgets, other hand-overs).  Every output is a codeword whose reported metric is its
    weighted Hamming distance; the conventional order-3 scan (all 43 745 TEPs) is never worse and equal where no rule fired."""
    from short_ldpc_decoding_osd_amd import _lib
    g = torch.Generator(device=dec.device).manual_seed(int(snr * 10))
    G = to_dev(dec.code.G, dec, torch.float32)
    Hm = to_dev(dec.code.H, dec, torch.float32)
    cw = (torch.randint(0, 2, (B, 64), device=dec.device, generator=g).to(torch.float32) @ G).remainder(2)
    sigma = np_oracle.snr_to_sigma(snr, 64, 128)
    y = ((1 - 2 * cw) * (1 + sigma * torch.randn((B, 128), device=dec.device, generator=g))).contiguous()
    res = dec.nms(y, 10, ALPHA0)
    index, count = dec.compact(res["fail"])
    nf = int(count.cpu()[0])
    idx = index[:nf].to(torch.int64)
    yf = y[idx].contiguous()

    def run(yy, n):
        aux = torch.zeros((n, 4), dtype=torch.int32, device=dec.device)
        out = dec.osd_decode(yy, 3, params=dec.osd_params(3, _lib.OSD_PB, snr_db=snr, aux=aux))
        torch.cuda.synchronize()
        return {k: out[k][:n].clone() for k in ("cw", "metric", "best", "ntep")} | {"aux": aux}

    a, b = run(yf, nf), run(yf, nf)
    for k in a:
        assert torch.equal(a[k], b[k]), k                                     # timing-independent
    step = 3000
    for s in range(0, nf, step * 7):                                          # (every seventh slice: a few dozen launches)
        e = min(s + step, nf)
        c = run(yf[s:e].contiguous(), e - s)
        for k in a:
            assert torch.equal(a[k][s:e], c[k]), (k, s)                       # batch-independent
    bits = dec.unpack_bits(a["cw"].contiguous()).to(torch.float32)
    assert not ((bits @ Hm.T).remainder(2) != 0).any()
    disc = (bits != (yf <= 0).to(torch.float32)).to(torch.float32)
    wsum = (disc.double() * yf.abs().double()).sum(1)
    assert torch.allclose(wsum, a["metric"].double(), rtol=1e-5, atol=1e-5)
    full = dec.osd_decode(yf, 3)
    torch.cuda.synchronize()
    assert (full["metric"][:nf] <= a["metric"]).all()
    nostop = a["aux"][:, 3] == 0
    assert (a["ntep"][nostop] == 43745).all() and (a["ntep"][~nostop] < 43745).all()
    assert torch.equal(full["metric"][:nf][nostop], a["metric"][nostop])
    assert a["ntep"].float().mean() < 43745 / 10
