"""Code definition: the call surface of the reference's ``fill_matrix_info.Code``
(LDPC_128/Ldpc_128_testing/fill_matrix_info.py:3-129), backed by libldpcosd.so.

    code = Code("CCSDS_ldpc_n128_k64.alist")
    code.H, code.G, code.k, code.check_matrix_row, code.check_matrix_column, code.max_chk_degree

Same attribute names, same int matrices (NumPy int64, as ``np.zeros(...).astype(int)`` gives
on Linux, :84), same alist dialect (:74-104) and the same G: reduced row-echelon form of H
with the reference's pivot rule, ``G = [H2^T | I]``, column exchanges undone (:44-69).
A failed ``H.G^T = 0`` check raises instead of printing (:63-68).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
CCSDS_128_64 = os.path.join(DATA_DIR, "CCSDS_ldpc_n128_k64.alist")


def _i32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class Code:
    def __init__(self, H_filename=None, H=None):
        self._handle = C.c_void_p()
        self._L = _lib.load()
        if H is not None:
            Hc = np.ascontiguousarray(H, dtype=np.int32)
            _lib.check(self._L.ldpc_code_from_dense(_i32p(Hc), Hc.shape[0], Hc.shape[1], C.byref(self._handle)),
                       "ldpc_code_from_dense")
        else:
            self.load_code(H_filename if H_filename is not None else CCSDS_128_64)
            return
        self._fill()

    def load_code(self, H_filename):
        """alist -> H, G, k ... (fill_matrix_info.py:70-129)"""
        path = H_filename
        if not os.path.exists(path) and os.path.exists(os.path.join(DATA_DIR, os.path.basename(path))):
            path = os.path.join(DATA_DIR, os.path.basename(path))  # shipped code definitions
        _lib.check(self._L.ldpc_code_from_alist(os.fsencode(path), C.byref(self._handle)), "ldpc_code_from_alist")
        self._fill()

    def _fill(self):
        n, m, k, mcd = (C.c_int32() for _ in range(4))
        _lib.check(self._L.ldpc_code_dims(self._handle, C.byref(n), C.byref(m), C.byref(k), C.byref(mcd)))
        H = np.empty((m.value, n.value), dtype=np.int32)
        G = np.empty((k.value, n.value), dtype=np.int32)
        _lib.check(self._L.ldpc_code_get_H(self._handle, _i32p(H)))
        _lib.check(self._L.ldpc_code_get_G(self._handle, _i32p(G)))
        self.H = H.astype(np.int64)
        self.G = G.astype(np.int64)
        self.max_chk_degree = mcd.value
        self.check_matrix_column = n.value
        self.check_matrix_row = m.value
        self.k = k.value

    def gf2elim(self, M):
        """``Code.gf2elim`` (:7-42): returns (reduced M, [(j, col), ...]).  Like the reference
        it works in place on an int matrix when no all-zero row has to be deleted."""
        Mi = np.ascontiguousarray(M, dtype=np.int32)
        m, n = Mi.shape
        swaps = np.zeros((n, 2), dtype=np.int32)
        ns, rows = C.c_int32(), C.c_int32()
        _lib.check(self._L.ldpc_gf2elim_host(_i32p(Mi), m, n, _i32p(swaps), C.byref(ns), C.byref(rows)),
                   "ldpc_gf2elim_host")
        out = Mi[: rows.value].astype(M.dtype if isinstance(M, np.ndarray) else np.int64)
        if isinstance(M, np.ndarray) and out.shape == M.shape:
            M[...] = out
            out = M
        return out, [(int(a), int(b)) for a, b in swaps[: ns.value]]

    def generator_matrix(self, parity_check_matrix):
        """``Code.generator_matrix`` (:44-69) for an arbitrary H."""
        return Code(H=parity_check_matrix).G

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value:
            self._L.ldpc_code_destroy(h)
            h.value = None
