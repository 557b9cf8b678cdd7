"""Test-frame generation and TFRecord writing with the reference's names
(LDPC_128/Testing_data_gen_128/data_generating.py:13-51, LDPC_128/Ldpc_128_testing/data_generating.py:8-26).
Host-side, not on the timed path: the benchmark generates its frames on the device."""
import numpy as np

from . import globalmap as GL
from .tfrecord import TFRecordWriter, encode_example, write_examples


def testing_data_generating(code, SNR, max_frame, rng=None):
    """AWGN frames: sigma = sqrt(1/(2 R 10^(SNR/10))), unit-mean channel, random message . G,
    BPSK 0 -> +1, NO 2/sigma^2 scaling (data_generating.py:13-51).  The reference draws from the
    unseeded global NumPy RNG (:10); pass ``rng`` (np.random.Generator) for reproducible sets."""
    n, k = code.check_matrix_column, code.k
    sigma = np.sqrt(1. / (2 * (float(k) / float(n)) * 10 ** (SNR / 10)))
    if GL.get_map('Rayleigh_fading', False):
        raise NotImplementedError("Rayleigh branch (data_generating.py:21-38) is out of scope")
    normal = rng.normal if rng is not None else np.random.normal
    integers = (lambda lo, hi, size: rng.integers(lo, hi, size=size)) if rng is not None else \
        (lambda lo, hi, size: np.random.randint(lo, hi, size=size))
    channel_information = normal(1, sigma, size=(max_frame, n))
    if not GL.get_map('ALL_ZEROS_CODEWORD_TESTING', False):
        rand_message = integers(0, 2, [max_frame, k])
        codewords = rand_message.dot(code.G) % 2
        testing_data = np.where(codewords == 0, channel_information, -channel_information)
        testing_data_labels = codewords.astype(np.int64)
    else:
        testing_data = channel_information
        testing_data_labels = np.zeros((max_frame, n), dtype=np.int64)
    return testing_data, testing_data_labels


def get_tfrecords_example(feature, label):
    return encode_example(feature, label)


def make_tfrecord(data, out_filename):
    """One Example per row (data_generating.py:16-26)."""
    feats, labels = data
    write_examples(out_filename, feats, labels)
