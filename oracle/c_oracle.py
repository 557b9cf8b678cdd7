"""ctypes face of oracle/ldpc_oracle.c -- TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the
product package.  ``build()`` (re)compiles the shared object with gcc via oracle/Makefile.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libldpc_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "ldpc_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(os.environ.get("LDPC_ORACLE_LIB") or _SO)     # (LDPC_ORACLE_LIB: the sanitizer build of build.py --asan)
        i32p, f32p, u8p, i64p = (C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_uint8),
                                 C.POINTER(C.c_int64))
        L.orc_gf2elim.argtypes = [i32p, C.c_int, C.c_int, i32p, i32p]
        L.orc_gf2elim.restype = C.c_int
        L.orc_generator.argtypes = [i32p, C.c_int, C.c_int, i32p]
        L.orc_generator.restype = C.c_int
        L.orc_nms.argtypes = [i32p, C.c_int, C.c_int, f32p, C.c_int64, C.c_int, f32p, C.c_float, C.c_float,
                              f32p, f32p]
        L.orc_nms.restype = None
        L.orc_eval.argtypes = [i32p, C.c_int, C.c_int, f32p, u8p, C.c_int64, u8p, u8p, i64p]
        L.orc_eval.restype = None
        L.orc_osd_front.argtypes = [i32p, C.c_int, C.c_int, f32p, i32p, i32p, i32p, i32p]
        L.orc_osd_front.restype = C.c_int
        L.orc_tep_table.argtypes = [C.c_int, C.c_int, u8p]
        L.orc_tep_table.restype = C.c_int64
        L.orc_conv_osd_batch.argtypes = [i32p, f32p, u8p, C.c_int64, C.c_int, i32p, f32p, u8p]
        L.orc_conv_osd_batch.restype = C.c_int
        L.orc_fs_osd_batch.argtypes = [i32p, f32p, u8p, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_float, i32p, f32p,
                                       u8p, u8p]
        L.orc_fs_osd_batch.restype = C.c_int
        L.orc_pb_osd_batch.argtypes = [i32p, f32p, u8p, C.c_int64, C.c_int, C.c_float, i32p, f32p, u8p]
        L.orc_pb_osd_batch.restype = C.c_int
        L.orc_det_expf.argtypes = [C.c_float]
        L.orc_det_expf.restype = C.c_float
        L.orc_binom_cdf64.argtypes = [C.c_double, C.POINTER(C.c_double)]
        L.orc_binom_cdf64.restype = None
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def gf2elim(M):
    M = _i32(M).copy()
    m, n = M.shape
    sw = np.zeros((n, 2), dtype=np.int32)
    ns = np.zeros(1, dtype=np.int32)
    rows = lib().orc_gf2elim(_p(M, C.c_int32), m, n, _p(sw, C.c_int32), _p(ns, C.c_int32))
    return M[:rows], [tuple(int(x) for x in r) for r in sw[: ns[0]]]


def generator(H):
    H = _i32(H)
    m, n = H.shape
    G = np.zeros((n, n), dtype=np.int32)
    k = lib().orc_generator(_p(H, C.c_int32), m, n, _p(G, C.c_int32))
    if k < 0:
        raise RuntimeError("H G^T != 0")
    return G.reshape(-1)[: k * n].reshape(k, n).copy()


def nms(H, y, T, alpha, w_in=1.0, w_out=1.0, want_traj=False):
    H = _i32(H)
    m, n = H.shape
    y = np.ascontiguousarray(y, dtype=np.float32)
    B = y.shape[0]
    alpha = np.ascontiguousarray(np.broadcast_to(np.asarray(alpha, dtype=np.float32), (max(T, 1),)))
    soft = np.empty((B, n), dtype=np.float32)
    traj = np.empty((T + 1, B, n), dtype=np.float32) if want_traj else None
    lib().orc_nms(_p(H, C.c_int32), m, n, _p(y, C.c_float), B, T, _p(alpha, C.c_float), w_in, w_out,
                  _p(traj, C.c_float), _p(soft, C.c_float))
    return (soft, traj) if want_traj else soft


def evaluate(H, soft, labels):
    H = _i32(H)
    m, n = H.shape
    soft = np.ascontiguousarray(soft, dtype=np.float32)
    B = soft.shape[0]
    lab = np.ascontiguousarray(labels, dtype=np.uint8) if labels is not None else None
    hard = np.empty((B, n), dtype=np.uint8)
    fail = np.empty(B, dtype=np.uint8)
    counts = np.zeros(5, dtype=np.int64)
    lib().orc_eval(_p(H, C.c_int32), m, n, _p(soft, C.c_float), _p(lab, C.c_uint8), B, _p(hard, C.c_uint8),
                   _p(fail, C.c_uint8), _p(counts, C.c_int64))
    return hard, fail, dict(zip(("frames", "frame_err", "bit_err", "undetected", "synd_fail"), counts.tolist()))


def osd_front(G, y):
    G = _i32(G)
    k, n = G.shape
    y = np.ascontiguousarray(y, dtype=np.float32)
    perm = np.empty(n, dtype=np.int32)
    Gp = np.empty((k, n), dtype=np.int32)
    sw = np.zeros((n, 2), dtype=np.int32)
    ns = np.zeros(1, dtype=np.int32)
    rc = lib().orc_osd_front(_p(G, C.c_int32), k, n, _p(y, C.c_float), _p(perm, C.c_int32), _p(Gp, C.c_int32),
                             _p(sw, C.c_int32), _p(ns, C.c_int32))
    if rc:
        raise RuntimeError("rank-deficient G")
    return perm, Gp, [tuple(int(x) for x in r) for r in sw[: ns[0]]]


def tep_table(k, order):
    nt = lib().orc_tep_table(k, order, None)
    t = np.empty((nt, 3), dtype=np.uint8)
    lib().orc_tep_table(k, order, _p(t, C.c_uint8))
    return t


def conv_osd(G, y, labels, order):
    """Conventional order-p OSD on [F,128] original-order frames.  Returns dict of arrays."""
    G = _i32(G)
    y = np.ascontiguousarray(y, dtype=np.float32)
    F = y.shape[0]
    lab = np.ascontiguousarray(labels, dtype=np.uint8) if labels is not None else None
    info = np.empty((F, 5), dtype=np.int32)
    metric = np.empty(F, dtype=np.float32)
    cw = np.empty((F, 128), dtype=np.uint8)
    rc = lib().orc_conv_osd_batch(_p(G, C.c_int32), _p(y, C.c_float), _p(lab, C.c_uint8), F, order,
                                  _p(info, C.c_int32), _p(metric, C.c_float), _p(cw, C.c_uint8))
    if rc:
        raise RuntimeError("orc_conv_osd_batch failed")
    return dict(best=info[:, 0].copy(), phase=info[:, 1].copy(), correct=info[:, 2].astype(bool),
                teps_size=int(info[0, 3]) if F else 0, nswaps=info[:, 4].copy(), metric=metric, codeword=cw)


def fs_osd(G, y, labels, order, beta=0.1, tau_e=6.5, tau_psc=30.0):
    """FS-OSD on [F,128] original-order frames (fs_testing.py:129-161).  Returns dict of arrays;
    ``*_ref`` follow the reference's ``optimal_codeword`` (quirk kept), ``*_hit`` the tau_e winner."""
    G = _i32(G)
    y = np.ascontiguousarray(y, dtype=np.float32)
    F = y.shape[0]
    lab = np.ascontiguousarray(labels, dtype=np.uint8) if labels is not None else None
    info = np.empty((F, 5), dtype=np.int32)
    met = np.empty((F, 2), dtype=np.float32)
    cw_ref = np.empty((F, 128), dtype=np.uint8)
    cw_hit = np.empty((F, 128), dtype=np.uint8)
    rc = lib().orc_fs_osd_batch(_p(G, C.c_int32), _p(y, C.c_float), _p(lab, C.c_uint8), F, order, beta, tau_e, tau_psc,
                                _p(info, C.c_int32), _p(met, C.c_float), _p(cw_ref, C.c_uint8), _p(cw_hit, C.c_uint8))
    if rc:
        raise RuntimeError("orc_fs_osd_batch failed")
    return dict(num_teps=info[:, 0].copy(), hit=info[:, 1].astype(bool), best_index=info[:, 2].copy(),
                correct_ref=info[:, 3].astype(bool), correct_hit=info[:, 4].astype(bool), metric_ref=met[:, 0].copy(),
                metric_hit=met[:, 1].copy(), codeword_ref=cw_ref, codeword_hit=cw_hit)


def pb_osd(G, y, labels, order, snr_db):
    """PB-OSD on [F,128] original-order frames (pb_testing.py:100-149)."""
    G = _i32(G)
    y = np.ascontiguousarray(y, dtype=np.float32)
    F = y.shape[0]
    lab = np.ascontiguousarray(labels, dtype=np.uint8) if labels is not None else None
    info = np.empty((F, 7), dtype=np.int32)
    metric = np.empty(F, dtype=np.float32)
    cw = np.empty((F, 128), dtype=np.uint8)
    rc = lib().orc_pb_osd_batch(_p(G, C.c_int32), _p(y, C.c_float), _p(lab, C.c_uint8), F, order, snr_db,
                                _p(info, C.c_int32), _p(metric, C.c_float), _p(cw, C.c_uint8))
    if rc:
        raise RuntimeError("orc_pb_osd_batch failed")
    return dict(num_teps=info[:, 0].copy(), best_index=info[:, 1].copy(), correct=info[:, 2].astype(bool),
                comparisons=info[:, 3].copy(), suc1=info[:, 4].copy(), suc2=info[:, 5].copy(), stop=info[:, 6].copy(),
                metric=metric, codeword=cw)


def det_expf(x):
    return np.array([lib().orc_det_expf(float(v)) for v in np.atleast_1d(x)], dtype=np.float32)


def binom_cdf64(p):
    out = np.empty(65, dtype=np.float64)
    lib().orc_binom_cdf64(float(p), out.ctypes.data_as(C.POINTER(C.c_double)))
    return out
