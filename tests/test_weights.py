"""CPU: values.txt weight import.  No trained file ships with the reference, so the
fixtures are written here in the exact format of ms_decoder_dense.py:338-346."""
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
