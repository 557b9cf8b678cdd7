"""CPU: TFRecord / tf.train.Example codec -- round trips, the reference's retest
layout ((T+1) rows per failed frame), checksum handling, and known CRC-32C / protobuf vectors."""
import struct

import numpy as np
import pytest

from short_ldpc_decoding_osd_amd import _lib, data_generating, read_TFdata, tfrecord


def test_crc32c_known_vectors():
    L = _lib.load()
    assert L.ldpc_crc32c(b"123456789", 9) == 0xE3069283            # CRC-32C check value
    assert L.ldpc_crc32c(b"", 0) == 0
    assert L.ldpc_crc32c(bytes(32), 32) == 0x8A9136AA               # RFC 3720 B.4: 32 bytes of zeros
    assert L.ldpc_crc32c(bytes([0xFF] * 32), 32) == 0x62A8AB43      # RFC 3720 B.4: 32 bytes of 0xFF


def test_example_wire_format_matches_protobuf_layout():
    ex = tfrecord.encode_example(np.array([1.0, -2.5], np.float32), np.array([0, 1]))
    # Example{features=1}{feature map entry=1}{key=1,value=2}{float_list=2|int64_list=3}{value=1 packed}
    assert ex[0] == 0x0A and b"\x0a\x07feature" in ex and b"\x0a\x05label" in ex and b"\x0a\x05shape" in ex
    assert struct.pack("<ff", 1.0, -2.5) in ex
    d = tfrecord.decode_example(ex)
    assert d["feature"].tolist() == [1.0, -2.5] and d["label"].tolist() == [0, 1] and d["shape"].tolist() == [2]
    # unpacked encodings (older writers) decode too
    unpacked = bytes([0x0A, 0x12, 0x0A, 0x10, 0x0A, 0x01, ord("x"), 0x12, 0x0B, 0x1A, 0x09, 0x08, 0x03, 0x08, 0x7F, 0x08]) + \
        bytes([0xFF] * 4)
    with pytest.raises(Exception):
        tfrecord.decode_example(unpacked)            # truncated varint must not be silently accepted


def test_round_trip_and_retest_layout(tmp_path):
    rng = np.random.default_rng(0)
    T = 3
    frames = 5
    feats = rng.normal(size=(frames * (T + 1), 128)).astype(np.float32)
    labels = np.repeat(rng.integers(0, 2, size=(frames, 128)), T + 1, axis=0).astype(np.int64)
    path = str(tmp_path / "ldpc-nonzero-retest.tfrecord")
    data_generating.make_tfrecord((feats, labels), path)
    # stage 6 reads with batch = unit_batch x (T+1) and uses row 0 (PB_OSD/globalmap.py:70, pb_testing.py:71)
    ds = read_TFdata.data_handler(128, path, 1 * (T + 1))
    got = list(ds.as_numpy_iterator())
    assert len(got) == frames
    for i, (f, lab, shp) in enumerate(got):
        assert f.shape == (T + 1, 128) and f.dtype == np.float32 and lab.dtype == np.int64
        assert np.array_equal(f, feats[i * (T + 1):(i + 1) * (T + 1)])
        assert np.array_equal(lab[0], labels[i * (T + 1)]) and (shp == 128).all()
    assert len(list(ds.take(2).as_numpy_iterator())) == 2
    # ragged tail batch is kept (drop_remainder=False, read_TFdata.py:27)
    tail = list(read_TFdata.data_handler(128, path, 7).as_numpy_iterator())
    assert [b[0].shape[0] for b in tail] == [7, 7, 6]


def test_corruption_is_detected(tmp_path):
    path = str(tmp_path / "x.tfrecord")
    data_generating.make_tfrecord((np.ones((2, 128), np.float32), np.zeros((2, 128), np.int64)), path)
    raw = bytearray(open(path, "rb").read())
    raw[40] ^= 0x01
    open(path, "wb").write(bytes(raw))
    with pytest.raises(IOError):
        list(tfrecord.read_records(path))
    open(path, "wb").write(bytes(raw[:100]))
    with pytest.raises(IOError):
        list(tfrecord.read_records(path))


def test_generator_statistics():
    from short_ldpc_decoding_osd_amd import Code
    code = Code()
    y, lab = data_generating.testing_data_generating(code, 2.5, 4000, rng=np.random.default_rng(1))
    assert y.shape == (4000, 128) and lab.shape == (4000, 128)
    assert not (lab.dot(code.H.T) % 2).any()                     # labels are codewords
    s = np.where(lab == 0, y, -y)
    assert abs(s.mean() - 1.0) < 0.01 and abs(s.std() - 0.749894) < 0.01     # sigma at 2.5 dB


def test_whole_file_paths_equal_the_record_by_record_codec(tmp_path):
    """The reference's files hold one kind of record; such a file is written and read as ONE byte matrix (round 4: the per-record
    Python codec was 99.9 % of the stage drivers' time).  Same bytes out, same arrays in; anything that does not fit the one
    layout -- labels beyond one varint byte, records of different lengths -- takes the record-by-record path."""
    rng = np.random.default_rng(3)
    N = 300
    feats = rng.normal(size=(N, 128)).astype(np.float32)
    labels = rng.integers(0, 2, size=(N, 128)).astype(np.int64)
    feats[7] = 1.0                                                # a row whose value bytes repeat
    p_bulk, p_rows = str(tmp_path / "bulk.tfrecord"), str(tmp_path / "rows.tfrecord")
    tfrecord.write_examples(p_bulk, feats, labels)
    with tfrecord.TFRecordWriter(p_rows) as w:
        for i in range(N):
            w.write(tfrecord.encode_example(feats[i], labels[i]))
    assert open(p_bulk, "rb").read() == open(p_rows, "rb").read()
    bulk = tfrecord.read_examples_bulk(p_rows, 128)
    assert bulk is not None and np.array_equal(bulk[0], feats) and np.array_equal(bulk[1], labels) and (bulk[2] == 128).all()
    rows = [tfrecord.decode_example(r) for r in tfrecord.read_records(p_rows)]
    assert all(np.array_equal(r["feature"], feats[i]) and np.array_equal(r["label"], labels[i]) for i, r in enumerate(rows))
    got = list(read_TFdata.data_handler(128, p_bulk, 64).as_numpy_iterator())
    assert [g[0].shape[0] for g in got] == [64, 64, 64, 64, 44] and np.array_equal(np.concatenate([g[1] for g in got]), labels)
    assert int(tfrecord.masked_crc_rows(np.frombuffer(b"123456789", np.uint8)[None, :])[0]) == tfrecord.masked_crc(b"123456789")
    # labels that need two varint bytes: no common layout -> the row path, for writer and reader
    big = labels.copy(); big[5, 9] = 300
    assert tfrecord.encode_examples_bulk(feats, big) is None
    p_big = str(tmp_path / "big.tfrecord")
    tfrecord.write_examples(p_big, feats, big)
    assert tfrecord.read_examples_bulk(p_big, 128) is None
    got = list(read_TFdata.data_handler(128, p_big, N).as_numpy_iterator())
    assert np.array_equal(got[0][1], big) and np.array_equal(got[0][0], feats)
    # a flipped byte inside the values of record 100: the whole-file reader reports it like the record reader does
    raw = bytearray(open(p_bulk, "rb").read())
    stride = len(raw) // N
    raw[100 * stride + 200] ^= 0x10
    open(p_bulk, "wb").write(bytes(raw))
    with pytest.raises(IOError):
        list(read_TFdata.data_handler(128, p_bulk, 64).as_numpy_iterator())
