"""MI355X-native short-LDPC decoder: normalised min-sum BP + ordered-statistics decoding.

Hot path = hand-written HIP kernels in csrc/ behind the C ABI of include/ldpc_osd.h
(libldpcosd.so, built in-tree by ``python -m short_ldpc_decoding_osd_amd.build``); the
modules here mirror the reference's Python call surface (same module / function names).
"""
from . import _lib  # noqa: F401
from .fill_matrix_info import Code  # noqa: F401

__all__ = ["Code"]
__version__ = "0.1.0"
