"""ctypes binding of libldpcosd.so (C ABI: include/ldpc_osd.h).

There is deliberately no fallback: if the shared object is missing or a call fails the
wrapper raises -- the decoders only exist as HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LDPC_OSD_LIB: another build of the same ABI (the host-only sanitizer build of `build.py --asan`, whose device entry
# points are stubs that fail)
LIB_PATH = os.environ.get("LDPC_OSD_LIB") or os.path.join(_HERE, "libldpcosd.so")

# every symbol include/ldpc_osd.h declares (tests check the list against the header)
SYMBOLS = (
    "ldpc_last_error", "ldpc_abi_version",
    "ldpc_code_from_alist", "ldpc_code_from_dense", "ldpc_code_destroy", "ldpc_code_dims",
    "ldpc_code_get_H", "ldpc_code_get_G", "ldpc_gf2elim_host", "ldpc_tep_table", "ldpc_tep_table_fs", "ldpc_crc32c",
    "ldpc_ctx_create", "ldpc_ctx_destroy", "ldpc_ctx_nms_kernel", "ldpc_ctx_get_pb_tuning", "ldpc_ctx_set_pb_tuning",
    "ldpc_nms_decode", "ldpc_nms_traj_rows", "ldpc_eval_counts", "ldpc_compact", "ldpc_pack_bits", "ldpc_unpack_bits",
    "ldpc_osd_reserve", "ldpc_osd_reserve_stream", "ldpc_osd_release_stream", "ldpc_osd_index_errors", "ldpc_osd_ge", "ldpc_osd_front", "ldpc_osd_search", "ldpc_osd_decode", "ldpc_osd_tep_eval", "ldpc_osd_counts",
    "ldpc_hosd_pattern_teps", "ldpc_hosd_front", "ldpc_hosd_search",
    "ldpc_pipeline_run", "ldpc_pipeline_timing",
)

NMS_AUTO, NMS_GENERIC, NMS_QC16 = 0, 1, 2
OSD_CONVENTIONAL, OSD_FS, OSD_PB = 0, 1, 2


class OsdParams(C.Structure):
    _fields_ = [("order", C.c_int32), ("algo", C.c_int32), ("snr_db", C.c_float), ("fs_beta", C.c_float),
                ("fs_tau_e", C.c_float), ("fs_tau_psc", C.c_float), ("fs_reference_quirk", C.c_int32),
                ("reserved", C.c_int32), ("d_aux", C.c_void_p), ("y_frames", C.c_int64)]


class PbTuning(C.Structure):
    """ldpc_pb_tuning of include/ldpc_osd.h (PB-OSD hand-over schedule and chunk targets of a context)."""
    _fields_ = [(n, C.c_int32) for n in ("budget", "budget_s", "budget_m", "budget_l", "budget_xl", "t1", "t2", "t3",
                                         "late_min", "late_maxlen", "late_pct", "late_div", "handoff_maxlen")]


class Pipeline(C.Structure):
    """ldpc_pipeline of include/ldpc_osd.h (device pointers as integers)."""
    _fields_ = [("d_llr", C.c_void_p), ("B", C.c_int64), ("T", C.c_int32), ("nms_kernel", C.c_int32),
                ("alpha", C.POINTER(C.c_float)), ("w_in", C.c_float), ("w_out", C.c_float),
                ("d_soft", C.c_void_p), ("d_hard", C.c_void_p), ("d_fail", C.c_void_p),
                ("d_label_bits", C.c_void_p), ("d_nms_counts", C.c_void_p),
                ("osd_enable", C.c_int32), ("timing_slot", C.c_int32), ("osd", OsdParams),
                ("d_index", C.c_void_p), ("d_count", C.c_void_p), ("d_perm", C.c_void_p), ("d_parity", C.c_void_p),
                ("d_cw", C.c_void_p), ("d_metric", C.c_void_p), ("d_best", C.c_void_p), ("d_ntep", C.c_void_p),
                ("d_osd_counts", C.c_void_p)]


TIMING_SLOTS = 64


class LdpcError(RuntimeError):
    pass


_lib = None


def load():
    """Load libldpcosd.so once; raise with build instructions if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LdpcError(
            f"{LIB_PATH} is missing: build it with `python -m short_ldpc_decoding_osd_amd.build` "
            "(hipcc, --offload-arch=gfx950). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    pi32, pi64 = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    sig = {
        "ldpc_last_error": (C.c_char_p, []),
        "ldpc_abi_version": (C.c_int, []),
        "ldpc_code_from_alist": (C.c_int, [C.c_char_p, C.POINTER(vp)]),
        "ldpc_code_from_dense": (C.c_int, [pi32, i32, i32, C.POINTER(vp)]),
        "ldpc_code_destroy": (None, [vp]),
        "ldpc_code_dims": (C.c_int, [vp, pi32, pi32, pi32, pi32]),
        "ldpc_code_get_H": (C.c_int, [vp, pi32]),
        "ldpc_code_get_G": (C.c_int, [vp, pi32]),
        "ldpc_gf2elim_host": (C.c_int, [pi32, i32, i32, pi32, pi32, pi32]),
        "ldpc_tep_table": (i64, [i32, i32, C.POINTER(C.c_uint8), pi64]),
        "ldpc_tep_table_fs": (i64, [i32, i32, C.POINTER(C.c_uint8)]),
        "ldpc_crc32c": (C.c_uint32, [vp, C.c_uint64]),
        "ldpc_ctx_create": (C.c_int, [vp, i32, C.POINTER(vp)]),
        "ldpc_ctx_destroy": (None, [vp]),
        "ldpc_ctx_nms_kernel": (C.c_int, [vp]),
        "ldpc_ctx_get_pb_tuning": (C.c_int, [vp, C.POINTER(PbTuning)]),
        "ldpc_ctx_set_pb_tuning": (C.c_int, [vp, C.POINTER(PbTuning)]),
        # device entry points: device pointers travel as integers (tensor.data_ptr())
        "ldpc_nms_decode": (C.c_int, [vp, vp, i64, i32, C.POINTER(f32), f32, f32, vp, vp, vp, vp, i32, vp]),
        "ldpc_nms_traj_rows": (C.c_int, [vp, vp, vp, vp, i64, i32, C.POINTER(f32), f32, f32, vp, i32, vp]),
        "ldpc_eval_counts": (C.c_int, [vp, vp, vp, vp, i64, vp, vp]),
        "ldpc_compact": (C.c_int, [vp, vp, i64, vp, vp, vp]),
        "ldpc_pack_bits": (C.c_int, [vp, vp, i32, i64, vp, vp]),
        "ldpc_unpack_bits": (C.c_int, [vp, vp, i64, vp, i32, vp]),
        "ldpc_osd_reserve": (C.c_int, [vp, i64]),
        "ldpc_osd_reserve_stream": (C.c_int, [vp, i64, C.POINTER(OsdParams), vp]),
        "ldpc_osd_release_stream": (C.c_int, [vp, vp]),
        "ldpc_osd_index_errors": (C.c_int, [vp, pi64]),
        "ldpc_osd_ge": (C.c_int, [vp, vp, i64, vp, vp, vp, vp]),
        "ldpc_osd_front": (C.c_int, [vp, vp, vp, vp, i64, vp, vp, vp, vp]),
        "ldpc_osd_search": (C.c_int, [vp, vp, vp, vp, i64, vp, vp, C.POINTER(OsdParams), vp, vp, vp, vp, vp]),
        "ldpc_osd_decode": (C.c_int, [vp, vp, vp, vp, i64, C.POINTER(OsdParams), vp, vp, vp, vp, vp]),
        "ldpc_osd_tep_eval": (C.c_int, [vp, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp]),
        "ldpc_osd_counts": (C.c_int, [vp, vp, vp, vp, vp, vp, i64, vp, vp]),
        "ldpc_hosd_pattern_teps": (i64, [i32, pi32, pi32, C.POINTER(C.c_uint8)]),
        "ldpc_hosd_front": (C.c_int, [vp, vp, i64, vp, vp, vp, vp, vp]),
        "ldpc_hosd_search": (C.c_int, [vp, vp, vp, i64, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]),
        "ldpc_pipeline_run": (C.c_int, [vp, C.POINTER(Pipeline), vp]),
        "ldpc_pipeline_timing": (C.c_int, [vp, i32, C.POINTER(f32)]),
    }
    for name in SYMBOLS:
        fn = getattr(L, name)  # AttributeError here = ABI/header drift, which must be loud
        fn.restype, fn.argtypes = sig[name]
    _lib = L
    return L


def check(rc, what=""):
    """Raise LdpcError for a negative status code."""
    if rc is not None and rc < 0:
        msg = load().ldpc_last_error()
        raise LdpcError(f"{what or 'libldpcosd'} failed ({rc}): {msg.decode() if msg else ''}")
    return rc
