"""-m gpu: the OSD kernels through the C ABI -- bit-exact against (i) the reference's own
gf2elim outputs (tests/golden/gf2elim_ccsds.npz) and (ii) the CPU oracle."""
import os

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


def _rows_packed(M):
    """[F,64,128] 0/1 -> [F,64,2] uint64 (bit c of row r at word c//64)."""
    F = M.shape[0]
    return np.packbits(M.astype(np.uint8), axis=2, bitorder="little").view(np.uint64).reshape(F, 64, 2)


def test_device_ge_matches_reference_gf2elim(dec, golden_dir):
    """ldpc_osd_ge vs the outputs of the reference's own Code.gf2elim (== full_gf2elim)."""
    g = np.load(os.path.join(golden_dir, "gf2elim_ccsds.npz"))
    G = dec.code.G
    perm = g["perm"].astype(np.int64)
    M = np.stack([G[:, p] for p in perm])
    want = _rows_packed(np.unpackbits(g["reduced"], axis=2))
    red, swaps, ns = dec.osd_ge(to_dev(_rows_packed(M).view(np.int64), dec))
    torch.cuda.synchronize()
    assert np.array_equal(ns.cpu().numpy(), g["nswaps"])
    assert np.array_equal(words_np(red).reshape(-1, 64, 2), want)
    sw = swaps.cpu().numpy()
    for i in range(M.shape[0]):
        n = int(g["nswaps"][i])
        assert np.array_equal(sw[i, :n], g["swaps"][i, :n]), i


def _front_oracle(dec, y):
    perm_o, par_o, ns_o = [], [], []
    for row in y:
        perm, Gp, sw = c_oracle.osd_front(dec.code.G, row)
        perm_o.append(perm)
        par_o.append(np.packbits(Gp[:, 64:].astype(np.uint8), axis=1, bitorder="little").view(np.uint64)[:, 0])
        ns_o.append(len(sw))
        assert np.array_equal(Gp[:, :64], np.eye(64, dtype=np.int32))
    return np.stack(perm_o), np.stack(par_o), np.array(ns_o)


def test_front_end_matches_oracle(dec, golden_dir):
    g = np.load(os.path.join(golden_dir, "gf2elim_ccsds.npz"))
    y = g["y"]
    perm, parity, ns = dec.osd_front(to_dev(y, dec))
    torch.cuda.synchronize()
    perm_o, par_o, ns_o = _front_oracle(dec, y)
    assert np.array_equal(ns.cpu().numpy(), ns_o)
    assert np.array_equal(perm.cpu().numpy(), perm_o)
    assert np.array_equal(words_np(parity), par_o)


def test_front_end_ties_and_zeros(dec):
    """Equal |y| (the case tf.argsort leaves open): the build's rule is 'lower index first'."""
    rng = np.random.default_rng(12)
    y, _ = np_oracle.make_frames(dec.code.G, 2.5, 48, rng)
    y[0] = 1.0                               # all equal
    y[1] = np.where(np.arange(128) % 2, -0.5, 0.5)
    y[2, :64] = 0.0                          # zeros (reliability 0) in the first half
    y[3] = np.round(y[3] * 4) / 4            # many duplicates
    y[4, 10] = -y[4, 20]
    y[5] = -0.0
    # other input scalings (the sort's value buckets scale with the frame's largest magnitude)
    y[6] *= 40.0
    y[7] *= 1e-3
    y[8, 77] = 1e30                          # one outlier: every other value lands in one bucket
    y[9, 5], y[9, 100] = np.inf, -np.inf
    y[10] *= 1e-41                           # denormals
    y[11, 3] = 3.0e38
    perm, parity, ns = dec.osd_front(to_dev(y, dec))
    torch.cuda.synchronize()
    perm_o, par_o, ns_o = _front_oracle(dec, y)
    assert np.array_equal(perm.cpu().numpy(), perm_o)
    assert np.array_equal(words_np(parity), par_o)
    assert np.array_equal(ns.cpu().numpy(), ns_o)


def test_front_end_saturated_inputs(dec):
    """Clipped / saturated values put most keys into ONE sort bucket (66 or more of the 128) while an exact zero sits in
    the lowest: the in-bucket count then runs 128 entries past a bucket start (it used to read past the
    64-entry pad and rank the zero at 128 -- a non-permutation).  Checked against the oracle's permutation."""
    rng = np.random.default_rng(77)
    y, _ = np_oracle.make_frames(dec.code.G, 2.5, 40, rng)
    y = np.clip(y, -1.0, 1.0)
    y[:, 5] = 0.0
    y[1, 127] = 0.0                           # the smallest possible key: (|0.0| bits, 127 - 127)
    y[2] = np.clip(y[2] * 8, -1.0, 1.0)       # nearly every value saturated
    y[2, 5] = 0.0
    y[3] = 1.0
    y[3, 127] = 0.0                           # 127 equal keys in the top bucket, one zero
    y[4] = np.where(rng.random(128) < 0.52, 1.0, y[4] * 0.01).astype(np.float32)
    perm, parity, ns = dec.osd_front(to_dev(y, dec))
    torch.cuda.synchronize()
    got = perm.cpu().numpy()
    assert all(sorted(p) == list(range(128)) for p in got.tolist())
    perm_o, par_o, ns_o = _front_oracle(dec, y)
    assert np.array_equal(got, perm_o)
    assert np.array_equal(words_np(parity), par_o)
    assert np.array_equal(ns.cpu().numpy(), ns_o)


@pytest.mark.parametrize("order", [0, 1, 2])
def test_conventional_osd_matches_oracle(dec, order):
    rng = np.random.default_rng(100 + order)
    y, cw = np_oracle.make_frames(dec.code.G, 2.5, 3000, rng)
    res = dec.nms(to_dev(y, dec), 10, ALPHA0)
    index, count = dec.compact(res["fail"])
    nf = int(count.cpu()[0])
    idx = index[:nf].cpu().numpy()
    assert nf > 500
    # device-side count + capacity form (no host round trip) and the explicit form must agree
    yd = to_dev(y, dec)
    out = dec.osd_decode(yd, order, index=index, count=count, F=y.shape[0])
    out2 = dec.osd_decode(yd, order, index=index[:nf].contiguous())
    torch.cuda.synchronize()
    ref = c_oracle.conv_osd(dec.code.G, y[idx], cw[idx], order)
    for o in (out, out2):
        assert np.array_equal(words_np(o["cw"])[:nf], pack_np(ref["codeword"]))
        assert np.array_equal(o["best"].cpu().numpy()[:nf], ref["best"])
        assert np.array_equal(o["metric"].cpu().numpy()[:nf], ref["metric"])
        assert (o["ntep"].cpu().numpy()[:nf] == ref["teps_size"]).all()
    label_bits = dec.pack_bits(to_dev(cw, dec))
    counts = dec.osd_counts(out["cw"], label_bits, index=index, count=count, ntep=out["ntep"], F=y.shape[0])
    c = counts.cpu().numpy()
    assert c[0] == nf and c[1] == int((~ref["correct"]).sum()) and c[2] == nf * ref["teps_size"]


def test_order3_and_direct_addressing(dec):
    rng = np.random.default_rng(5)
    y, cw = np_oracle.make_frames(dec.code.G, 2.0, 24, rng)
    out = dec.osd_decode(to_dev(y, dec), 3)          # no index: every frame, order 3 (43 745 TEPs)
    torch.cuda.synchronize()
    ref = c_oracle.conv_osd(dec.code.G, y, cw, 3)
    assert ref["teps_size"] == 43745
    assert np.array_equal(words_np(out["cw"]), pack_np(ref["codeword"]))
    assert np.array_equal(out["best"].cpu().numpy(), ref["best"])
    assert np.array_equal(out["metric"].cpu().numpy(), ref["metric"])


def test_osd_edge_cases(dec):
    from short_ldpc_decoding_osd_amd import _lib
    y = torch.zeros((0, 128), dtype=torch.float32, device=dec.device)
    out = dec.osd_decode(y, 2)
    assert out["cw"].shape == (0, 2)
    yy = to_dev(np.ones((4, 128), np.float32), dec)
    with pytest.raises(_lib.LdpcError):
        dec.osd_decode(yy, 4)
    with pytest.raises(_lib.LdpcError):
        dec.osd_decode(yy, 2, algo=7)
    # count = 0 on the device: nothing is written
    cnt = torch.zeros(1, dtype=torch.int32, device=dec.device)
    idx = torch.zeros(4, dtype=torch.int32, device=dec.device)
    sentinel = torch.full((4, 2), -1, dtype=torch.int64, device=dec.device)
    dec.osd_decode(yy, 2, index=idx, count=cnt, F=4, out=dict(cw=sentinel))
    torch.cuda.synchronize()
    assert (sentinel == -1).all()


def test_full_size_osd_properties(dec):
    """BASELINE config 3/4 size per GPU: size-independent properties of OSD-2 on the failures of
    131 072 frames (about 33 k OSD frames)."""
    B = 131072
    g = torch.Generator(device=dec.device).manual_seed(99)
    G = to_dev(dec.code.G, dec, torch.float32)
    Hm = to_dev(dec.code.H, dec, torch.float32)
    cw = (torch.randint(0, 2, (B, 64), device=dec.device, generator=g).to(torch.float32) @ G).remainder(2)
    sigma = np_oracle.snr_to_sigma(2.5, 64, 128)
    y = ((1 - 2 * cw) * (1 + sigma * torch.randn((B, 128), device=dec.device, generator=g))).contiguous()
    res = dec.nms(y, 10, ALPHA0)
    index, count = dec.compact(res["fail"])
    nf = int(count.cpu()[0])
    out0 = dec.osd_decode(y, 0, index=index, count=count, F=B)
    out2 = dec.osd_decode(y, 2, index=index, count=count, F=B)
    torch.cuda.synchronize()
    idx = index[:nf].to(torch.int64)
    bits2 = dec.unpack_bits(out2["cw"][:nf].contiguous()).to(torch.float32)
    assert not ((bits2 @ Hm.T).remainder(2) != 0).any()                   # every output is a codeword
    # the reported metric is the weighted Hamming distance to the hard decision of the channel
    ysel = y[idx]
    disc = (bits2 != (ysel <= 0).to(torch.float32)).to(torch.float32)
    wsum = (disc.double() * ysel.abs().double()).sum(1)
    assert torch.allclose(wsum, out2["metric"][:nf].double(), rtol=1e-5, atol=1e-5)
    assert (out2["metric"][:nf] <= out0["metric"][:nf]).all()              # more TEPs never hurt
    assert (out0["best"][:nf] == 0).all() and (out2["best"][:nf] < 2081).all()
    label_bits = dec.pack_bits(cw.to(torch.int64))
    c = dec.osd_counts(out2["cw"], label_bits, index=index, count=count, ntep=out2["ntep"], F=B).cpu().numpy()
    assert c[0] == nf and c[2] == nf * 2081
    fail_rate = c[1] / nf
    assert 0.03 < fail_rate < 0.09            # OSD-2 fails on ~5.8 % of the NMS failures
    # ML property: the true codeword is never strictly better than a correct OSD answer
    true_metric = (((cw[idx] != (ysel <= 0).to(torch.float32)).double()) * ysel.abs().double()).sum(1)
    assert (out2["metric"][:nf].double() <= true_metric * (1 + 1e-5) + 1e-5)[bits2.eq(cw[idx]).all(1)].all()


def test_order2_register_kernel_equals_table_scan_with_ties(dec):
    """The register-resident order-2 kernel against the table-driven scan and the oracle on inputs
    where many TEPs have EQUAL metrics (quantised |y|), so 'first minimum in table order' matters."""
    rng = np.random.default_rng(44)
    y, cw = np_oracle.make_frames(dec.code.G, 2.0, 600, rng)
    y[:200] = np.sign(y[:200]) * np.maximum(np.round(np.abs(y[:200]) * 2) / 2, 0.5)   # |y| in {0.5, 1, 1.5, ...}
    y[200:300] = np.sign(y[200:300])                                                   # all |y| = 1
    y[300:320] = np.where(rng.random((20, 128)) < 0.5, 1.0, -1.0).astype(np.float32) * 0.25
    yd = to_dev(y, dec)
    a = dec.osd_decode(yd, 2)                                                # rotation-paired persistent kernel
    b = dec.osd_decode(yd, 2, params=dec.osd_params(2, table_scan=True))
    c = dec.osd_decode(yd, 2, params=dec.osd_params(2, readlane_scan=True))  # first register-resident kernel
    torch.cuda.synchronize()
    for k in ("cw", "metric", "best", "ntep"):
        assert torch.equal(a[k], b[k]), k
        assert torch.equal(a[k], c[k]), k
    ref = c_oracle.conv_osd(dec.code.G, y, cw, 2)
    assert np.array_equal(a["best"].cpu().numpy(), ref["best"])
    assert np.array_equal(words_np(a["cw"]), pack_np(ref["codeword"]))
    assert np.array_equal(a["metric"].cpu().numpy(), ref["metric"])


def test_single_call_pipeline_equals_the_separate_calls(dec):
    """ldpc_pipeline_run == ldpc_nms_decode + eval + compact + osd_front + osd_search + osd_counts."""
    from short_ldpc_decoding_osd_amd.pipeline import BatchPipeline
    rng = np.random.default_rng(71)
    y, cw = np_oracle.make_frames(dec.code.G, 2.5, 5000, rng)
    yd, lab = to_dev(y, dec), dec.pack_bits(to_dev(cw, dec))
    pipe = BatchPipeline(dec, 5000, 10, ALPHA0, osd_order=2).bind(yd, lab)                       # own front-end buffers
    pipe.run(timing_slot=3)
    pipe.run()
    pipe_ws = BatchPipeline(dec, 5000, 10, ALPHA0, osd_order=2, keep_front=False).bind(yd, lab)   # context workspace
    pipe_ws.run(timing_slot=4)
    torch.cuda.synchronize()
    nf0 = int(pipe.count.cpu()[0])
    assert int(pipe_ws.count.cpu()[0]) == nf0 and torch.equal(pipe.index[:nf0], pipe_ws.index[:nf0])
    for k in ("cw", "metric", "best", "ntep"):
        assert torch.equal(getattr(pipe, k)[:nf0], getattr(pipe_ws, k)[:nf0]), k
    perm_o, par_o, _ = dec.osd_front(yd, index=pipe.index, count=pipe.count, F=5000)
    torch.cuda.synchronize()
    assert torch.equal(pipe.perm[:nf0], perm_o[:nf0]) and torch.equal(pipe.parity[:nf0], par_o[:nf0])
    assert pipe.timing(3)[1] > 2 * pipe_ws.timing(4)[1]        # workspace path: no separate front-end interval
    res = dec.nms(yd, 10, ALPHA0)
    index, count = dec.compact(res["fail"])
    nf = int(count.cpu()[0])
    out = dec.osd_decode(yd, 2, index=index, count=count, F=5000)
    torch.cuda.synchronize()
    assert torch.equal(pipe.soft, res["soft"]) and torch.equal(pipe.hard, res["hard"]) and torch.equal(pipe.fail, res["fail"])
    assert int(pipe.count.cpu()[0]) == nf and torch.equal(pipe.index[:nf], index[:nf])
    for k, t in (("cw", pipe.cw), ("metric", pipe.metric), ("best", pipe.best), ("ntep", pipe.ntep)):
        assert torch.equal(t[:nf], out[k][:nf]), k
    c = pipe.counters().cpu().numpy()
    ref = c_oracle.conv_osd(dec.code.G, y[index[:nf].cpu().numpy()], cw[index[:nf].cpu().numpy()], 2)
    assert c[0] == 10000 and c[4] == 2 * nf and c[5] == 2 * nf and c[6] == 2 * int((~ref["correct"]).sum())
    ms = pipe.timing(3)
    assert all(0.0 < v < 50.0 for v in ms)
    # NMS-only pipeline and a PB-OSD pipeline through the same entry point
    from short_ldpc_decoding_osd_amd import _lib
    p2 = BatchPipeline(dec, 5000, 10, ALPHA0).bind(yd, lab)
    p2.run()
    p3 = BatchPipeline(dec, 5000, 10, ALPHA0, osd_order=2, osd_algo=_lib.OSD_PB, snr_db=2.5).bind(yd, lab)
    p3.run()
    torch.cuda.synchronize()
    assert torch.equal(p2.soft, res["soft"]) and p2.counters().cpu().numpy()[4] == nf
    refpb = c_oracle.pb_osd(dec.code.G, y[index[:nf].cpu().numpy()], cw[index[:nf].cpu().numpy()], 2, 2.5)
    assert np.array_equal(p3.ntep[:nf].cpu().numpy(), refpb["num_teps"])
    assert p3.counters().cpu().numpy()[6] == int((~refpb["correct"]).sum())


def test_tep_eval_matches_oracle(dec):
    """ldpc_osd_tep_eval (one_tep_compare, fs_testing.py:51-64): a GIVEN error pattern of any weight per frame, on the
    device front end's results, against the NumPy restatement (re-encode, Hamming and canonical weighted distance)."""
    rng = np.random.default_rng(77)
    y, cw = np_oracle.make_frames(dec.code.G, 2.0, 300, rng)
    yd = to_dev(y, dec)
    perm, parity, _ = dec.osd_front(yd)
    weights = rng.integers(0, 7, size=300)
    masks = np.zeros(300, dtype=np.uint64)
    for f in range(300):
        for p in rng.choice(64, size=weights[f], replace=False):
            masks[f] |= np.uint64(1) << np.uint64(p)
    out = dec.osd_tep_eval(yd, perm, parity, to_dev(masks.view(np.int64), dec))
    torch.cuda.synchronize()
    got_cw, got_m, got_hd = words_np(out["cw"]), out["metric"].cpu().numpy(), out["hd"].cpu().numpy()
    for f in range(0, 300, 3):
        yp, lp, Gp, pm, _ = np_oracle.swapped_info(y[f], cw[f], dec.code.G)
        assert np.array_equal(perm[f].cpu().numpy(), pm)
        hard = np.where(yp > 0, 0, 1).astype(np.int64)
        e = np.array([(int(masks[f]) >> p) & 1 for p in range(64)], dtype=np.int64)
        cand = ((hard[:64] + e) % 2).dot(Gp) % 2
        disc = (cand + hard) % 2
        assert got_hd[f] == disc.sum()
        assert got_m[f] == np_oracle.weighted_distance(disc, np.abs(yp))
        orig = np.empty(128, dtype=np.int64)
        orig[pm] = cand
        assert np.array_equal(got_cw[f], pack_np(orig[None])[0])
