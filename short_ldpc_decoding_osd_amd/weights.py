"""Learned-weight import.

The training stage mirrors its variables to ``ckpts/<type>/<T>th/par/values.txt``
(LDPC_128/Ldpc_128_training/ms_decoder_dense.py:338-346): a header line
``For all layers at the %4d-th step:`` followed by ``<variable name> <numpy str>`` for every model
variable, written back to back without separators, then a newline.  Values are the STORED
(pre-softplus) weights; the decoder applies softplus (ms_test.py:207-208).
"""
from __future__ import annotations

import re

import numpy as np



def softplus32(x):
    """softplus of a stored weight -> effective factor (ms_test.py:207-208 applies tf.nn.softplus to a float32
    variable).  ONE convention for the whole package: log1p(exp(x)) evaluated in float32.  (TensorFlow is not
    available to pin the last bit -- parity unpinned; for the shipped weight -0.048 the float32 and the float64
    evaluation round to the same float32, 0.66943514.)"""
    x = np.float32(x)
    return np.float32(np.log1p(np.exp(x)))


STORED_NMS1_WEIGHT = -0.048      # Decoder_Layer's initial value, the only weight the reference ships (ms_test.py:73,83)

_HEADER = re.compile(r"For all layers at the\s*(\d+)-th step:")
_ENTRY = re.compile(r"([A-Za-z_][\w /.\-]*?(?::\d+)?)\s*\[\s*([-+0-9.eE]+(?:\s+[-+0-9.eE]+)*)\s*\]")

# add_weight names in Decoder_Layer.build (ms_test.py:82-91) -> attribute names
_ATTR = (
    ("decoder_bit_normalized factor1", "shared_bit_weight1"),
    ("decoder_bit_normalized factor2", "shared_bit_weight2"),
    ("decoder_bit_normalized factor", "shared_bit_weight"),
    ("decoder_check_normalized factor", "shared_check_weight"),
)


def parse_values_txt(path, step="latest"):
    """-> (step, {variable name: np.float32 array}) of the requested record (default: the last)."""
    text = open(path, "rt").read()
    heads = list(_HEADER.finditer(text))
    if not heads:
        raise ValueError(f"{path}: no 'For all layers at the N-th step:' record")
    if step == "latest":
        pick = len(heads) - 1
    else:
        wanted = [i for i, h in enumerate(heads) if int(h.group(1)) == int(step)]
        if not wanted:
            raise KeyError(f"{path}: no record for step {step}")
        pick = wanted[-1]
    body = text[heads[pick].end(): heads[pick + 1].start() if pick + 1 < len(heads) else len(text)]
    out = {}
    for m in _ENTRY.finditer(body):
        out[m.group(1).strip()] = np.array([float(v) for v in m.group(2).split()], dtype=np.float32)
    if not out:
        raise ValueError(f"{path}: record for step {heads[pick].group(1)} holds no variables")
    return int(heads[pick].group(1)), out


def apply_to_layer(layer, variables):
    """Copy parsed variables into a ``ms_test.Decoder_Layer`` (stored values, as in the checkpoint)."""
    done = []
    for name, value in variables.items():
        base = name.split("/")[-1].split(":")[0]
        for key, attr in _ATTR:
            if base == key:
                setattr(layer, attr, np.asarray(value, dtype=np.float32).reshape(-1)[:1].copy())
                done.append(attr)
                break
    if "shared_check_weight" not in done:
        raise KeyError("values.txt record has no 'decoder_check_normalized factor' variable")
    return done


def load_values_txt(model, path, step="latest"):
    """Restore ``Decoding_model`` weights from a values.txt mirror; returns the step loaded."""
    step_loaded, variables = parse_values_txt(path, step)
    apply_to_layer(model.layer, variables)
    return step_loaded
