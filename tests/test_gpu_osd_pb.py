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


@pytest.mark.parametrize("snr,order,frames", [(2.5, 2, 3000), (2.5, 3, 1500), (1.0, 2, 600), (3.5, 3, 6000), (2.5, 1, 800)])
def test_pb_matches_oracle(dec, snr, order, frames):
    from short_ldpc_decoding_osd_amd import _lib
    y, cw = _failures(dec, snr, frames, seed=int(snr * 10) + order)
    y, cw = y[:600], cw[:600]
    ref = c_oracle.pb_osd(dec.code.G, y, cw, order, snr)
    aux = torch.zeros((y.shape[0], 4), dtype=torch.int32, device=dec.device)
    p = dec.osd_params(order, _lib.OSD_PB, snr_db=snr, aux=aux)
    out = dec.osd_decode(to_dev(y, dec), order, params=p)
    torch.cuda.synchronize()
    a = aux.cpu().numpy()
    assert np.array_equal(out["ntep"].cpu().numpy(), ref["num_teps"])
    assert np.array_equal(a[:, 3], ref["stop"])
    assert np.array_equal(a[:, 0], ref["comparisons"]) and np.array_equal(a[:, 1], ref["suc1"])
    assert np.array_equal(a[:, 2], ref["suc2"])
    assert np.array_equal(out["best"].cpu().numpy(), ref["best_index"])
    assert np.array_equal(words_np(out["cw"]), pack_np(ref["codeword"]))
    assert np.array_equal(out["metric"].cpu().numpy(), ref["metric"])
    # SURVEY 6 scale check: PB-OSD visits ~1e2 TEPs per frame, far below the 2081 / 43745 of the full scan
    assert ref["num_teps"].mean() < 1000


def test_pb_full_scan_and_spill(dec):
    """A frame on which no rule fires runs all N_max - 1 TEPs (frontier spills past the LDS head)."""
    from short_ldpc_decoding_osd_amd import _lib
    y, cw = _failures(dec, 2.5, 1500, seed=8)
    y, cw = y[:64], cw[:64]
    # snr_db = 40 dB makes every bit-error probability tiny: the success rule needs a near-perfect match
    for snr in (-20.0, 40.0):
        ref = c_oracle.pb_osd(dec.code.G, y, cw, 2, snr)
        p = dec.osd_params(2, _lib.OSD_PB, snr_db=snr)
        out = dec.osd_decode(to_dev(y, dec), 2, params=p)
        torch.cuda.synchronize()
        assert np.array_equal(out["ntep"].cpu().numpy(), ref["num_teps"])
        assert np.array_equal(words_np(out["cw"]), pack_np(ref["codeword"]))
        assert np.array_equal(out["metric"].cpu().numpy(), ref["metric"])
