"""-m gpu: randomised end-to-end parity -- NMS-T + conventional OSD-p through the one-call pipeline against the C
oracle, over random SNR / iteration count / scale factor / batch size / order draws (fixed seeds).  Everything
is compared bit for bit: posteriors, hard words, syndrome flags, the failure list, OSD winners and metrics."""
import numpy as np
import pytest
import torch

from oracle import c_oracle, np_oracle
from tests.gpu_util import pack_np, to_dev, words_np

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dec():
    from short_ldpc_decoding_osd_amd import Code
    from short_ldpc_decoding_osd_amd.runtime import Decoder
    return Decoder(Code())


@pytest.mark.parametrize("seed", range(10))
def test_random_configuration_matches_oracle(dec, seed):
    from short_ldpc_decoding_osd_amd.pipeline import BatchPipeline
    rng = np.random.default_rng(1000 + seed)
    snr = float(rng.choice([0.5, 1.5, 2.0, 2.5, 3.0, 4.0]))
    T = int(rng.choice([1, 3, 7, 10, 12, 16]))
    alpha = np.float32(rng.uniform(0.4, 1.0))
    order = int(rng.choice([0, 1, 2, 2, 2]))
    B = int(rng.integers(1, 1400))
    y, cw = np_oracle.make_frames(dec.code.G, snr, B, rng)
    if seed % 3 == 0:                                            # sprinkle exact zeros and ties
        y[rng.integers(0, B, 5), rng.integers(0, 128, 5)] = 0.0
        r = int(rng.integers(0, B))
        y[r, 40:44] = y[r, 7]
    pipe = BatchPipeline(dec, B, T, alpha, osd_order=order).bind(to_dev(y, dec), dec.pack_bits(to_dev(cw, dec)))
    pipe.run()
    torch.cuda.synchronize()
    soft = c_oracle.nms(dec.code.H, y, T, alpha)
    hard, fail, counts = c_oracle.evaluate(dec.code.H, soft, cw)
    assert np.array_equal(pipe.soft.cpu().numpy(), soft)
    assert np.array_equal(words_np(pipe.hard), pack_np(hard)) and np.array_equal(pipe.fail.cpu().numpy(), fail)
    idx = np.flatnonzero(fail)
    nf = int(pipe.count.cpu()[0])
    assert nf == len(idx) and np.array_equal(pipe.index[:nf].cpu().numpy(), idx)
    c = pipe.counters().cpu().numpy()
    assert [int(v) for v in c[:5]] == [counts[k] for k in ("frames", "frame_err", "bit_err", "undetected", "synd_fail")]
    if nf:
        ref = c_oracle.conv_osd(dec.code.G, y[idx], cw[idx], order)
        assert np.array_equal(pipe.best[:nf].cpu().numpy(), ref["best"])
        assert np.array_equal(pipe.metric[:nf].cpu().numpy(), ref["metric"])
        assert np.array_equal(words_np(pipe.cw[:nf]), pack_np(ref["codeword"]))
        assert int(c[5]) == nf and int(c[6]) == int((~ref["correct"]).sum())
