"""CPU: values.txt weight import.  No trained file ships with the reference, so the
fixtures are written here in the exact format of ms_decoder_dense.py:338-346."""
import os

import numpy as np
import pytest

from short_ldpc_decoding_osd_amd import weights


def _write(path, records):
    with open(path, "a+") as f:
        for step, variables in records:
            f.write("For all layers at the %4d-th step:\n" % step)
            for name, val in variables:
                f.write(name + ' ' + str(np.array([val], dtype=np.float32)))
            f.write('\n')


def test_parse_latest_and_named_step(tmp_path):
    p = str(tmp_path / "values.txt")
    _write(p, [(50, [("decoder_check_normalized factor:0", -0.048)]),
               (100, [("decoder__layer/decoder_check_normalized factor:0", -0.3125)])])
    step, v = weights.parse_values_txt(p)
    assert step == 100 and list(v.values())[0][0] == np.float32(-0.3125)
    step, v = weights.parse_values_txt(p, step=50)
    assert step == 50 and list(v.values())[0][0] == np.float32(-0.048)
    with pytest.raises(KeyError):
        weights.parse_values_txt(p, step=75)


def test_multi_variable_records_and_layer_update(tmp_path):
    p = str(tmp_path / "values.txt")
    _write(p, [(200, [("decoder_bit_normalized factor1:0", 0.25), ("decoder_bit_normalized factor2:0", -1.5e-3),
                      ("decoder_check_normalized factor:0", 0.5)])])

    class Layer:
        pass

    layer = Layer()
    _, v = weights.parse_values_txt(p)
    done = weights.apply_to_layer(layer, v)
    assert sorted(done) == ["shared_bit_weight1", "shared_bit_weight2", "shared_check_weight"]
    assert layer.shared_check_weight[0] == np.float32(0.5) and layer.shared_bit_weight2[0] == np.float32(-1.5e-3)
    bad = str(tmp_path / "empty.txt")
    open(bad, "w").write("nothing here\n")
    with pytest.raises(ValueError):
        weights.parse_values_txt(bad)


# ---- TensorFlow checkpoint bundles (SURVEY 8(f) N3 ii).  No TF-written file exists to test against (the
# reference ships none, TensorFlow is absent): these are round trips through the module's own writer, which
# lays the bytes out as TensorFlow's tensor_bundle / table sources describe -- "parity unpinned".
def test_checkpoint_bundle_round_trip(tmp_path):
    from short_ldpc_decoding_osd_amd import tf_checkpoint as ck
    key = "myAwesomeModel/layer/shared_check_weight" + ck.VARIABLE_SUFFIX
    tensors = {key: np.array([-0.3125], dtype=np.float32),
               "myAwesomeModel/layer/other/.ATTRIBUTES/VARIABLE_VALUE": np.arange(12, dtype=np.int64).reshape(3, 4),
               "save_counter/.ATTRIBUTES/VARIABLE_VALUE": np.array(7, dtype=np.int64),
               "half": np.array([1.5, -2.0], dtype=np.float16)}
    prefix = ck.write_checkpoint(str(tmp_path / "ckpts" / "ldpc-ckpt-100"), tensors)
    assert ck.latest_checkpoint(str(tmp_path / "ckpts")) == prefix
    header, entries = ck.read_index(prefix + ".index")
    assert header["num_shards"] == 1 and set(entries) == set(tensors)
    assert entries[key]["shape"] == (1,) and entries[key]["dtype"] == 1 and entries[key]["size"] == 4
    back = ck.read_checkpoint(prefix)
    for k, v in tensors.items():
        assert back[k].dtype == v.dtype and back[k].shape == v.shape and np.array_equal(back[k], v), k


def test_checkpoint_restores_the_decoder_weight_and_detects_damage(tmp_path):
    from short_ldpc_decoding_osd_amd import tf_checkpoint as ck

    class Layer:
        pass

    class Model:
        layer = Layer()

    d = tmp_path / "NMS-1" / "10th"
    prefix = ck.write_checkpoint(str(d / "ldpc-ckpt-250"), {
        "myAwesomeModel/layer/shared_check_weight" + ck.VARIABLE_SUFFIX: np.array([0.4321], dtype=np.float32)})
    m = Model()
    assert ck.load_checkpoint(m, str(d)) == ["shared_check_weight"]           # directory -> latest checkpoint
    assert m.layer.shared_check_weight[0] == np.float32(0.4321)
    assert ck.load_checkpoint(Model(), prefix) == ["shared_check_weight"]     # explicit prefix
    raw = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    raw[1] ^= 0x40
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="checksum"):
        ck.read_checkpoint(prefix)
    idx = bytearray(open(prefix + ".index", "rb").read())
    idx[5] ^= 0x01
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(ValueError, match="checksum"):
        ck.read_index(prefix + ".index")
    open(prefix + ".index", "wb").write(b"not a table at all, but at least forty-eight bytes long ....")
    with pytest.raises(ValueError, match="magic"):
        ck.read_index(prefix + ".index")
    with pytest.raises(FileNotFoundError):
        ck.load_checkpoint(Model(), str(tmp_path))
    other = ck.write_checkpoint(str(tmp_path / "x" / "c-1"), {"unrelated" + ck.VARIABLE_SUFFIX: np.zeros(1, np.float32)})
    with pytest.raises(KeyError):
        ck.load_checkpoint(Model(), other)


# ---- reader robustness against what a real TensorFlow writer emits and the module's own writer does not:
# LevelDB-style tables with prefix-compressed keys (restart interval 16), several data blocks, more than one data
# shard, and a compressed block (must be refused with a clear message).  The fixtures are built here byte by byte
# from the published table format (tensorflow/core/lib/io/table_builder.cc, block_builder.cc, format.cc;
# tensor_bundle.proto) -- still no TensorFlow-written file: parity unpinned.
def _table(rows, restart_interval=16, block_size=200, tag=0):
    import struct
    from short_ldpc_decoding_osd_amd.tfrecord import _varint, masked_crc

    def block(entries):
        body, restarts, last, n = bytearray(), [], b"", 0
        for key, value in entries:
            shared = 0
            if n % restart_interval == 0:
                restarts.append(len(body))
            else:
                while shared < min(len(key), len(last)) and key[shared] == last[shared]:
                    shared += 1
            body += _varint(shared) + _varint(len(key) - shared) + _varint(len(value)) + key[shared:] + value
            last, n = key, n + 1
        for r in restarts or [0]:
            body += struct.pack("<I", r)
        body += struct.pack("<I", max(len(restarts), 1))
        tagged = bytes(body) + bytes([tag])
        return tagged + struct.pack("<I", masked_crc(tagged)), len(body)

    out, index, cur = bytearray(), [], []
    size = 0
    for row in rows + [None]:
        if row is not None:
            cur.append(row)
            size += len(row[0]) + len(row[1])
        if cur and (row is None or size >= block_size):
            blk, n = block(cur)
            index.append((cur[-1][0] + b"\x00", _varint(len(out)) + _varint(n)))     # a separator key >= the block's last key
            out += blk
            cur, size = [], 0
    moff = len(out); mb, mn = block([]); out += mb
    ioff = len(out); ib, isz = block(index); out += ib
    footer = _varint(moff) + _varint(mn) + _varint(ioff) + _varint(isz)
    return bytes(out + footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57))


def _bundle_rows(tensors, shard_of, nshards):
    import struct
    from short_ldpc_decoding_osd_amd import tf_checkpoint as ck
    from short_ldpc_decoding_osd_amd.tfrecord import _len_field, _varint, masked_crc
    shards = [bytearray() for _ in range(nshards)]
    rows = [(b"", _varint((1 << 3) | 0) + _varint(nshards) + _len_field(3, _varint((1 << 3) | 0) + _varint(1)))]
    for name in sorted(tensors):
        a = np.asarray(tensors[name])
        raw, sid = a.tobytes(), shard_of(name)
        shape = b"".join(_len_field(2, _varint((1 << 3) | 0) + _varint(int(d))) for d in a.shape)
        entry = (_varint((1 << 3) | 0) + _varint(ck._DTYPE_IDS[np.dtype(a.dtype).str]) + _len_field(2, shape)
                 + _varint((3 << 3) | 0) + _varint(sid) + _varint((4 << 3) | 0) + _varint(len(shards[sid]))
                 + _varint((5 << 3) | 0) + _varint(len(raw)) + _varint((6 << 3) | 5) + struct.pack("<I", masked_crc(raw)))
        rows.append((name.encode(), entry))
        shards[sid] += raw
    return rows, shards


def test_checkpoint_reader_handles_tensorflow_style_tables(tmp_path):
    from short_ldpc_decoding_osd_amd import tf_checkpoint as ck
    rng = np.random.default_rng(1)
    tensors = {f"myAwesomeModel/layer/w{i:03d}" + ck.VARIABLE_SUFFIX: rng.standard_normal(3 + i % 4).astype(np.float32) for i in range(60)}
    tensors["myAwesomeModel/layer/shared_check_weight" + ck.VARIABLE_SUFFIX] = np.array([0.25], dtype=np.float32)
    tensors["save_counter" + ck.VARIABLE_SUFFIX] = np.array(3, dtype=np.int64)
    rows, shards = _bundle_rows(tensors, lambda name: sum(name.encode()) % 2 if "w0" in name else 0, 2)
    prefix = str(tmp_path / "ldpc-ckpt-9")
    open(prefix + ".index", "wb").write(_table(rows))                      # prefix-compressed keys, ~20 data blocks
    for sid, blob in enumerate(shards):
        open(f"{prefix}.data-{sid:05d}-of-00002", "wb").write(bytes(blob))
    header, entries = ck.read_index(prefix + ".index")
    assert header["num_shards"] == 2 and set(entries) == set(tensors)
    assert {e["shard_id"] for e in entries.values()} == {0, 1}
    back = ck.read_checkpoint(prefix)
    for k, v in tensors.items():
        assert back[k].dtype == v.dtype and np.array_equal(back[k], v), k

    class Layer:
        pass

    class Model:
        layer = Layer()

    assert ck.load_checkpoint(Model(), prefix) == ["shared_check_weight"] and Model.layer.shared_check_weight[0] == np.float32(0.25)
    # every restart interval gives the same entries
    for interval in (1, 2, 7, 64):
        open(prefix + ".index", "wb").write(_table(rows, restart_interval=interval, block_size=1 << 20))     # one big block
        assert set(ck.read_index(prefix + ".index")[1]) == set(tensors)
    # a snappy-compressed block (tag 1) is refused, not mis-parsed
    open(prefix + ".index", "wb").write(_table(rows, tag=1))
    with pytest.raises(ValueError, match="compressed"):
        ck.read_index(prefix + ".index")
    # a missing shard file is a clear error
    open(prefix + ".index", "wb").write(_table(rows))
    os.remove(f"{prefix}.data-00001-of-00002")
    with pytest.raises(FileNotFoundError):
        ck.read_checkpoint(prefix)
