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
