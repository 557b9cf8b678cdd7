"""Conventional order-p OSD with the reference's surface (LDPC_128/FS_OSD/convention_osd.py:13-76;
the PB_OSD copy unpacks six values where its caller passes five, :50 vs pb_testing.py:83 -- the
FS_OSD copy is the working one and is what is mirrored)."""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _lib
from . import globalmap as GL
from ._osd_common import primed_search

_ORDER_BY_ROWS = {1: 0, 65: 1, 2081: 2, 43745: 3}


def binomial_coefficient(n, k):
    return math.factorial(n) // (math.factorial(k) * math.factorial(n - k))


def _supports(k, order_limit):
    L = _lib.load()
    nt = _lib.check(L.ldpc_tep_table(k, order_limit, None, None), "ldpc_tep_table")
    t = np.empty((nt, 3), dtype=np.uint8)
    L.ldpc_tep_table(k, order_limit, t.ctypes.data_as(C.POINTER(C.c_uint8)), None)
    return t


def generate_binary_arrays(length, hamming_weight):
    """All weight-w patterns of ``length`` bits ordered as the reference orders them (:13-26)."""
    if hamming_weight < 0 or hamming_weight > length:
        return []
    lo = sum(binomial_coefficient(length, w) for w in range(hamming_weight))
    sup = _supports(length, hamming_weight)[lo:]
    out = np.zeros((sup.shape[0], length), dtype=np.int32)
    for q in range(hamming_weight):
        out[np.arange(sup.shape[0]), sup[:, q]] = 1
    return list(out)


def generate_teps(order_limit):
    """-> int32 [N, k] TEP matrix, weight classes 0..order_limit concatenated (:31-38)."""
    code = GL.get_map('code_parameters')
    sup = _supports(code.k, order_limit)
    out = np.zeros((sup.shape[0], code.k), dtype=np.int32)
    for q in range(3):
        rows = np.flatnonzero(sup[:, q] != 255)
        out[rows, sup[rows, q]] = 1
    return out


def query_boundary(order_limit):
    code = GL.get_map('code_parameters')
    L = _lib.load()
    b = (C.c_int64 * 4)()
    _lib.check(L.ldpc_tep_table(code.k, order_limit, None, b), "ldpc_tep_table")
    return [int(x) for x in list(b)[: order_limit + 1]]


def convention_osd_main(wrapped_input):
    """(updated_inputs, updated_labels, reduced_G, error_patterns_matrix, boundary_list) ->
    (correct_indicator, teps_size, belonged_phase)  (:49-76).  The TEP matrix must be the one
    ``generate_teps`` returns (its row count selects the order); the search runs on the device.
    The winning codeword / metric / index are kept in ``convention_osd_main.last``."""
    updated_inputs, updated_labels, reduced_G, error_patterns_matrix, boundary_list = wrapped_input
    rows = int(np.asarray(error_patterns_matrix).shape[0])
    if rows not in _ORDER_BY_ROWS:
        raise ValueError(f"TEP matrix with {rows} rows is not a generate_teps() table (1, 65, 2081 or 43745 rows)")
    order = _ORDER_BY_ROWS[rows]
    res = primed_search(updated_inputs, reduced_G, order, _lib.OSD_CONVENTIONAL)
    estimated_index = int(res["best"][0])
    correct_indicator = bool(np.array_equal(res["codeword"], np.asarray(updated_labels).astype(np.int32)))
    belonged_phase = -1
    if correct_indicator:
        for i in range(len(boundary_list)):
            if estimated_index < boundary_list[i]:
                belonged_phase = i
                break
    convention_osd_main.last = dict(codeword=res["codeword"], metric=float(res["metric"][0]), index=estimated_index)
    return correct_indicator, rows, belonged_phase
