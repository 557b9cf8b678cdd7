"""-m gpu: the HIP NMS kernels (through the C ABI) against the CPU oracle -- bit-exact on
soft outputs, trajectories, hard words and syndrome flags -- plus the small streaming
kernels (eval counters, compaction, bit packing)."""
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle, np_oracle
from tests.gpu_util import pack_np, to_dev, words_np

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALPHA0 = 0.669435


@pytest.fixture(scope="module")
def dec():
    from short_ldpc_decoding_osd_amd import Code
    from short_ldpc_decoding_osd_amd.runtime import Decoder
    return Decoder(Code())


@pytest.fixture(scope="module")
def H(dec):
    return dec.code.H


def _frames(dec, snr, B, seed):
    rng = np.random.default_rng(seed)
    return np_oracle.make_frames(dec.code.G, snr, B, rng)


def _check_against_oracle(dec, y, T, alpha, w_in=1.0, w_out=1.0, kernel=0):
    Hm = dec.code.H
    soft_o, traj_o = c_oracle.nms(Hm, y, T, alpha, w_in, w_out, want_traj=True)
    hard_o, fail_o, _ = c_oracle.evaluate(Hm, soft_o, None)
    res = dec.nms(to_dev(y, dec), T, alpha, w_in, w_out, want_traj=True, kernel=kernel)
    torch.cuda.synchronize()
    assert np.array_equal(res["soft"].cpu().numpy(), soft_o)
    if T:
        assert np.array_equal(res["traj"].cpu().numpy(), traj_o[1:])
    assert np.array_equal(words_np(res["hard"]), pack_np(hard_o))
    assert np.array_equal(res["fail"].cpu().numpy(), fail_o)
    return res


def test_library_selects_qc16_for_ccsds(dec):
    from short_ldpc_decoding_osd_amd import _lib
    assert dec.nms_kernel == _lib.NMS_QC16


@pytest.mark.parametrize("kernel", [1, 2], ids=["generic", "qc16"])
@pytest.mark.parametrize("snr", [1.0, 2.5, 3.5])
@pytest.mark.parametrize("T,alpha", [(1, 1.0), (10, ALPHA0), (12, 0.8)])
def test_nms_bit_exact(dec, kernel, snr, T, alpha):
    y, _ = _frames(dec, snr, 1000, seed=int(snr * 100) + T)
    _check_against_oracle(dec, y, T, alpha, kernel=kernel)


@pytest.mark.parametrize("kernel", [1, 2], ids=["generic", "qc16"])
def test_nms_per_iteration_alpha_and_bit_weights(dec, kernel):
    y, _ = _frames(dec, 2.0, 333, seed=9)
    alpha = np.linspace(0.5, 1.0, 7).astype(np.float32)
    _check_against_oracle(dec, y, 7, alpha, kernel=kernel)                    # learned per-iteration factors
    _check_against_oracle(dec, y, 7, alpha, 0.9, 0.9, kernel=kernel)          # NMS-2 (:127-128, :222-223)
    _check_against_oracle(dec, y, 7, alpha, 0.8, 1.1, kernel=kernel)          # NMS-3 (:129-130, :224-225)


@pytest.mark.parametrize("kernel", [1, 2], ids=["generic", "qc16"])
@pytest.mark.parametrize("B", [1, 2, 3, 5, 17, 63, 65])
def test_nms_ragged_batches(dec, kernel, B):
    y, _ = _frames(dec, 2.5, B, seed=B)
    _check_against_oracle(dec, y, 10, ALPHA0, kernel=kernel)


@pytest.mark.parametrize("kernel", [1, 2], ids=["generic", "qc16"])
def test_nms_zero_llr_and_zero_iterations(dec, kernel):
    y, _ = _frames(dec, 2.0, 64, seed=4)
    y[0, :5] = 0.0      # sign(0) = 0 wipes whole check rows (ms_test.py:187)
    y[1, 3] = -0.0
    y[2] = 0.0          # every bit decides to 1 (ms_test.py:39)
    y[3, ::2] = 0.0
    y[4] = 1e-30
    y[5] = -3e29
    _check_against_oracle(dec, y, 10, ALPHA0, kernel=kernel)
    _check_against_oracle(dec, y, 0, ALPHA0, kernel=kernel)


def test_nms_empty_batch(dec):
    res = dec.nms(torch.empty((0, 128), dtype=torch.float32, device=dec.device), 10, ALPHA0)
    assert res["soft"].shape == (0, 128) and res["fail"].shape == (0,)


@pytest.mark.parametrize("name,alist", [
    ("array_121_60", "tests/golden/ArrayCode_N121_K60_r0.50.alist"),
    ("ldpc_96_48", "tests/golden/LDPC_N96_K48_P8_set0_dmin10.alist"),
    ("wimax_1056", "tests/golden/wimax_1056_0.83.alist")])           # 80 KiB of LDS per block: opt-in path
def test_generic_kernel_other_codes(name, alist):
    from short_ldpc_decoding_osd_amd import Code, _lib
    from short_ldpc_decoding_osd_amd.runtime import Decoder
    d = Decoder(Code(os.path.join(ROOT, alist)))
    assert d.nms_kernel == _lib.NMS_GENERIC
    rng = np.random.default_rng(3)
    y, _ = np_oracle.make_frames(d.code.G, 3.0, 200, rng)
    _check_against_oracle(d, y, 8, 0.75)
    with pytest.raises(_lib.LdpcError):
        d.nms(to_dev(y, d), 8, 0.75, kernel=_lib.NMS_QC16)
    # the failed-frame rows (ldpc_nms_traj_rows) on this code's shape: rows of the listed frames = rows of the full trajectory
    yd = to_dev(y, d)
    res = d.nms(yd, 8, 0.75, want_traj=True)
    index, count = d.compact(res["fail"])
    nf = int(count.cpu()[0])
    if nf:
        rows = d.nms_traj_rows(yd, index, count, nf, 8, 0.75)
        torch.cuda.synchronize()
        idx = index[:nf].long()
        assert torch.equal(rows, torch.cat([yd[idx].unsqueeze(1), res["traj"][:, idx, :].permute(1, 0, 2)], dim=1))


def test_eval_counts_and_compaction(dec):
    y, cw = _frames(dec, 2.0, 5000, seed=21)
    res = dec.nms(to_dev(y, dec), 10, ALPHA0)
    soft = res["soft"].cpu().numpy()
    _, fail_o, cnt_o = c_oracle.evaluate(dec.code.H, soft, cw)
    label_bits = dec.pack_bits(to_dev(cw, dec))
    assert np.array_equal(words_np(label_bits), pack_np(cw))
    for dt in (torch.uint8, torch.int32):
        assert torch.equal(dec.pack_bits(to_dev(cw, dec, dt)), label_bits)
    assert np.array_equal(dec.unpack_bits(label_bits).cpu().numpy(), cw)
    counts = dec.eval_counts(res["hard"], label_bits, res["fail"]).cpu().numpy()
    assert dict(zip(("frames", "frame_err", "bit_err", "undetected", "synd_fail"), counts.tolist())) == cnt_o
    index, count = dec.compact(res["fail"])
    nf = int(count.cpu()[0])
    assert np.array_equal(index[:nf].cpu().numpy(), np.flatnonzero(fail_o))


@pytest.mark.parametrize("B", [0, 1, 7, 2047, 2048, 2049, 100003])
@pytest.mark.parametrize("p", [0.0, 0.25, 1.0])
def test_compaction_sizes(dec, B, p):
    rng = np.random.default_rng(B + int(p * 10))
    flag = (rng.random(B) < p).astype(np.uint8) * rng.integers(1, 255, size=B, dtype=np.uint8)
    index, count = dec.compact(to_dev(flag, dec))
    nf = int(count.cpu()[0])
    want = np.flatnonzero(flag)
    assert nf == want.size and np.array_equal(index[:nf].cpu().numpy(), want)


def test_full_size_properties(dec):
    """BASELINE config 2 size (65 536 frames): size-independent checks instead of the oracle."""
    B = 65536
    g = torch.Generator(device=dec.device).manual_seed(20241020)
    G = to_dev(dec.code.G, dec, torch.float32)
    msg = torch.randint(0, 2, (B, 64), device=dec.device, generator=g).to(torch.float32)
    cw = (msg @ G).remainder(2)
    sigma = np_oracle.snr_to_sigma(2.5, 64, 128)
    y = ((1 - 2 * cw) * (1 + sigma * torch.randn((B, 128), device=dec.device, generator=g))).contiguous()
    a = dec.nms(y, 10, ALPHA0, kernel=2)
    b = dec.nms(y, 10, ALPHA0, kernel=1)
    for k in ("soft", "hard", "fail"):
        assert torch.equal(a[k], b[k]), k                       # two independent kernels agree bit for bit
    hard = dec.unpack_bits(a["hard"]).to(torch.float32)
    assert torch.equal(hard, (a["soft"] <= 0).to(torch.float32))          # hard = (soft > 0 ? 0 : 1)
    synd = (hard @ to_dev(dec.code.H, dec, torch.float32).T).remainder(2).sum(1)
    assert torch.equal((synd != 0).to(torch.uint8), a["fail"])   # flag == non-zero syndrome
    counts = dec.eval_counts(a["hard"], dec.pack_bits(cw.to(torch.int64)), a["fail"]).cpu().numpy()
    fer = counts[1] / B
    assert counts[0] == B and 0.20 < fer < 0.31                   # SURVEY probe: ~0.25 at 2.5 dB
    assert counts[4] <= counts[1] and counts[3] == counts[1] - counts[4]
    # linearity of the code under the decoder's symmetry: flipping the sign pattern of a
    # codeword maps outputs by the same pattern
    flip = (1 - 2 * cw[:4096])
    c = dec.nms((y[:4096] * flip).contiguous(), 10, ALPHA0)
    assert torch.equal(c["soft"] * flip, a["soft"][:4096])


def test_decoding_model_surface(dec):
    """The reference-style objects (ms_test.py:26-70) on top of the same kernels."""
    from short_ldpc_decoding_osd_amd import globalmap as GL
    from short_ldpc_decoding_osd_amd import ms_test
    GL.set_map('code_parameters', dec.code)
    GL.set_map('num_iterations', 10)
    GL.set_map('selected_decoder_type', 'NMS-1')
    model = ms_test.Decoding_model()
    assert model.layer.num_iterations == 10
    y, cw = _frames(dec, 2.5, 300, seed=77)
    fer, ber, und, (bi, bl) = model(y, cw)
    alpha = np_oracle.softplus(-0.048)
    outs = np_oracle.nms_dense(y, dec.code.H, 10, alpha)
    fer_o, ber_o, und_o, idx = np_oracle.evaluate(outs[-1], cw, dec.code.H)
    assert fer == pytest.approx(fer_o) and float(ber) == pytest.approx(ber_o) and und == und_o
    rows, labs = np_oracle.collect_failed(outs, cw, idx)
    assert len(bi) == len(rows) == 11 * len(idx)
    assert np.array_equal(np.stack(bi), np.stack(rows)) and np.array_equal(np.stack(bl), np.stack(labs))
    lst = model.layer(y, cw)
    assert len(lst) == 11 and all(np.array_equal(a, b) for a, b in zip(lst, outs))
    f2, b2, u2, index = model.get_eval(lst, cw)
    assert (f2, u2) == (fer, und) and np.array_equal(index[:, 0], idx)


@pytest.mark.parametrize("kernel", ["auto", "generic"])
def test_traj_rows_of_listed_frames(dec, kernel):
    """ldpc_nms_traj_rows (collect_failed_output_selective, ms_test.py:55-64): T + 1 rows per LISTED frame, row 0 the channel
    values, row t the posterior after iteration t -- equal to the rows of the full [T][B][n] trajectory bit for bit, for both
    NMS kernels; the number of frames is device data (count), the capacity may be larger, an empty list writes nothing."""
    from short_ldpc_decoding_osd_amd import _lib
    kid = {"auto": _lib.NMS_AUTO, "generic": _lib.NMS_GENERIC}[kernel]
    T, alpha = 10, np_oracle.softplus(-0.048)
    y, cw = _frames(dec, 2.5, 1003, seed=31)
    yd = torch.from_numpy(y).to(dec.device)
    res = dec.nms(yd, T, alpha, want_traj=True, kernel=kid)
    index, count = dec.compact(res["fail"])
    nf = int(count.cpu()[0])
    assert 100 < nf < 600
    rows = dec.nms_traj_rows(yd, index, count, nf + 37, T, alpha, kernel=kid, out=torch.full((nf + 37, T + 1, 128), -7.0, device=dec.device))
    torch.cuda.synchronize()
    idx = index[:nf].long()
    want = torch.cat([yd[idx].unsqueeze(1), res["traj"][:, idx, :].permute(1, 0, 2)], dim=1)
    assert torch.equal(rows[:nf], want)
    assert (rows[nf:] == -7.0).all()                                       # capacity beyond the count: untouched
    # the C oracle's trajectory (soft outputs per iteration) for a few of them
    soft_o = c_oracle.nms(dec.code.H, y[idx[:8].cpu().numpy()], T, alpha)
    assert np.array_equal(rows[:8, T].cpu().numpy(), soft_o)
    # a caller-made list in another order, with a repeated frame; NMS-3 style bit weights
    lst = torch.tensor([5, 0, 1002, 5], dtype=torch.int32, device=dec.device)
    cnt = torch.tensor([4], dtype=torch.int32, device=dec.device)
    r2 = dec.nms_traj_rows(yd, lst, cnt, 4, 3, [0.5, 0.6, 0.7], w_in=0.9, w_out=1.1, kernel=kid)
    full = dec.nms(yd, 3, [0.5, 0.6, 0.7], w_in=0.9, w_out=1.1, want_traj=True, kernel=kid)["traj"]
    torch.cuda.synchronize()
    for k, f in enumerate([5, 0, 1002, 5]):
        assert torch.equal(r2[k, 0], yd[f]) and torch.equal(r2[k, 1:], full[:, f, :])
    zero = torch.zeros(1, dtype=torch.int32, device=dec.device)
    r3 = dec.nms_traj_rows(yd, lst, zero, 4, 3, 0.5, kernel=kid, out=torch.full((4, 4, 128), 3.0, device=dec.device))
    torch.cuda.synchronize()
    assert (r3 == 3.0).all()
