"""Device runtime: a thin object around ``ldpc_ctx`` that takes torch tensors as device
buffers and hands their pointers to the C ABI (include/ldpc_osd.h).  PyTorch is plumbing
here (allocation, streams, torch.distributed) -- all arithmetic happens in the HIP kernels.

Packed bit words are carried as ``torch.int64`` tensors (bit v of a frame = bit v%64 of
word v//64); view them as uint64 on the NumPy side.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .fill_matrix_info import Code

COUNT_NAMES = ("frames", "frame_err", "bit_err", "undetected", "synd_fail")


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class Decoder:
    """One per (code, GPU).  Every method is asynchronous on torch's current stream."""

    def __init__(self, code: Code | None = None, device=None):
        self.L = _lib.load()
        self.code = code if code is not None else Code()
        if not torch.cuda.is_available():
            raise _lib.LdpcError("no GPU visible to torch: the decoder has no CPU path")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else
                                   (device.index if isinstance(device, torch.device) else int(device)))
        self._ctx = C.c_void_p()
        _lib.check(self.L.ldpc_ctx_create(self.code._handle, self.device.index, C.byref(self._ctx)), "ldpc_ctx_create")
        self.n = self.code.check_matrix_column
        self.m = self.code.check_matrix_row
        self.k = self.code.k
        self.words = (self.n + 63) // 64
        self.nms_kernel = self.L.ldpc_ctx_nms_kernel(self._ctx)

    def __del__(self):
        ctx = getattr(self, "_ctx", None)
        if ctx is not None and ctx.value:
            self.L.ldpc_ctx_destroy(ctx)
            ctx.value = None

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _chk(self, t, dtype, shape_tail, name):
        if not isinstance(t, torch.Tensor) or t.device != self.device:
            raise ValueError(f"{name}: expected a tensor on {self.device}")
        if t.dtype != dtype or not t.is_contiguous():
            raise ValueError(f"{name}: expected contiguous {dtype}, got {t.dtype} contiguous={t.is_contiguous()}")
        if tuple(t.shape[1:]) != tuple(shape_tail):
            raise ValueError(f"{name}: expected shape [*, {shape_tail}], got {tuple(t.shape)}")
        return t

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    # ------------------------------------------------------------------ NMS
    def nms(self, llr, T, alpha, w_in=1.0, w_out=1.0, want_soft=True, want_traj=False, want_hard=True,
            want_fail=True, kernel=_lib.NMS_AUTO, out=None):
        """Normalised min-sum, T fixed iterations.  Returns dict(soft, traj, hard, fail) of
        tensors (None for outputs not requested).  ``out`` may carry preallocated tensors."""
        self._chk(llr, torch.float32, (self.n,), "llr")
        B = llr.shape[0]
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(alpha, dtype=np.float32), (max(T, 1),)))
        out = dict(out or {})
        if want_soft and out.get("soft") is None:
            out["soft"] = self.empty((B, self.n), torch.float32)
        if want_traj and out.get("traj") is None:
            out["traj"] = self.empty((T, B, self.n), torch.float32)
        if want_hard and out.get("hard") is None:
            out["hard"] = self.empty((B, self.words), torch.int64)
        if want_fail and out.get("fail") is None:
            out["fail"] = self.empty((B,), torch.uint8)
        for k in ("soft", "traj", "hard", "fail"):
            out.setdefault(k, None)
        _lib.check(self.L.ldpc_nms_decode(self._ctx, _ptr(llr), B, T, a.ctypes.data_as(C.POINTER(C.c_float)),
                                          float(w_in), float(w_out), _ptr(out["soft"]), _ptr(out["traj"]),
                                          _ptr(out["hard"]), _ptr(out["fail"]), int(kernel), self._stream()),
                   "ldpc_nms_decode")
        return out

    def nms_traj_rows(self, llr, index, count, F, T, alpha, w_in=1.0, w_out=1.0, kernel=_lib.NMS_AUTO, out=None):
        """Rows of the listed frames only (collect_failed_output_selective, ms_test.py:55-64): [F, T+1, n] f32, row 0 the
        channel values, row t the posterior after iteration t.  ``index`` / ``count``: the list as ``compact`` wrote it;
        ``F``: how many rows to allocate for (the launch decodes min(count, F) frames)."""
        self._chk(llr, torch.float32, (self.n,), "llr")
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(alpha, dtype=np.float32), (max(T, 1),)))
        rows = out if out is not None else self.empty((max(int(F), 1), T + 1, self.n), torch.float32)
        _lib.check(self.L.ldpc_nms_traj_rows(self._ctx, _ptr(llr), _ptr(index), _ptr(count), int(F), T,
                                             a.ctypes.data_as(C.POINTER(C.c_float)), float(w_in), float(w_out), _ptr(rows),
                                             int(kernel), self._stream()), "ldpc_nms_traj_rows")
        return rows

    # ------------------------------------------------------------------ statistics / plumbing kernels
    def eval_counts(self, hard, label_bits, fail=None, counts=None):
        """counts[5] += {frames, frame_err, bit_err, undetected, synd_fail} (int64 tensor)."""
        self._chk(hard, torch.int64, (self.words,), "hard")
        self._chk(label_bits, torch.int64, (self.words,), "label_bits")
        if counts is None:
            counts = torch.zeros(5, dtype=torch.int64, device=self.device)
        _lib.check(self.L.ldpc_eval_counts(self._ctx, _ptr(hard), _ptr(label_bits), _ptr(fail), hard.shape[0],
                                           _ptr(counts), self._stream()), "ldpc_eval_counts")
        return counts

    def compact(self, flag, index=None, count=None):
        """Ascending indices of the non-zero flags.  Returns (index[B] int32, count[1] int32)."""
        self._chk(flag, torch.uint8, (), "flag")
        B = flag.shape[0]
        if index is None:
            index = self.empty((max(B, 1),), torch.int32)
        if count is None:
            count = self.empty((1,), torch.int32)
        _lib.check(self.L.ldpc_compact(self._ctx, _ptr(flag), B, _ptr(index), _ptr(count), self._stream()),
                   "ldpc_compact")
        return index, count

    def pack_bits(self, bits):
        """[B, n] 0/1 tensor (uint8 / int32 / int64) -> [B, words] packed int64."""
        es = {torch.uint8: 1, torch.int32: 4, torch.int64: 8}.get(bits.dtype)
        if es is None:
            raise ValueError(f"pack_bits: unsupported dtype {bits.dtype}")
        self._chk(bits, bits.dtype, (self.n,), "bits")
        words = self.empty((bits.shape[0], self.words), torch.int64)
        _lib.check(self.L.ldpc_pack_bits(self._ctx, _ptr(bits), es, bits.shape[0], _ptr(words), self._stream()),
                   "ldpc_pack_bits")
        return words

    def unpack_bits(self, words, dtype=torch.int64):
        es = {torch.uint8: 1, torch.int32: 4, torch.int64: 8}[dtype]
        self._chk(words, torch.int64, (self.words,), "words")
        bits = self.empty((words.shape[0], self.n), dtype)
        _lib.check(self.L.ldpc_unpack_bits(self._ctx, _ptr(words), words.shape[0], _ptr(bits), es, self._stream()),
                   "ldpc_unpack_bits")
        return bits

    # ------------------------------------------------------------------ OSD
    def osd_reserve(self, max_frames):
        """Pre-size the OSD workspace (needed before capturing decode calls into a graph)."""
        _lib.check(self.L.ldpc_osd_reserve(self._ctx, int(max_frames)), "ldpc_osd_reserve")

    def osd_reserve_stream(self, max_frames, params=None):
        """Pre-size the OSD workspace of the CURRENT stream (instead of one eager call on it before a capture).
        ``params`` (osd_params(...)): also size what that search needs (PB-OSD lists and tables)."""
        _lib.check(self.L.ldpc_osd_reserve_stream(self._ctx, int(max_frames), C.byref(params) if params is not None else None,
                                                  self._stream()), "ldpc_osd_reserve_stream")

    def osd_release_stream(self, stream=None):
        """Free the OSD workspace of ``stream`` (default: the current stream).  The stream must be idle and graphs
        captured on it must not be replayed afterwards (include/ldpc_osd.h, 'Captured graphs')."""
        st = self._stream() if stream is None else C.c_void_p(stream.cuda_stream)
        _lib.check(self.L.ldpc_osd_release_stream(self._ctx, st), "ldpc_osd_release_stream")

    def pb_tuning(self):
        """The context's PB-OSD tuning (hand-over budgets, chunk targets) as a dict; see include/ldpc_osd.h."""
        t = _lib.PbTuning()
        _lib.check(self.L.ldpc_ctx_get_pb_tuning(self._ctx, C.byref(t)), "ldpc_ctx_get_pb_tuning")
        return {n: int(getattr(t, n)) for n, _ in _lib.PbTuning._fields_}

    def set_pb_tuning(self, **fields):
        """Change PB-OSD tuning fields of the context (no field: back to the defaults).  Results do not depend on them;
        applies to calls issued afterwards.  Returns the previous setting (a dict that can be passed back)."""
        prev = self.pb_tuning()
        if not fields:
            _lib.check(self.L.ldpc_ctx_set_pb_tuning(self._ctx, None), "ldpc_ctx_set_pb_tuning")
            return prev
        unknown = set(fields) - set(prev)
        if unknown:
            raise ValueError(f"unknown PB-OSD tuning field(s): {sorted(unknown)}")
        t = _lib.PbTuning(**{**prev, **{k: int(v) for k, v in fields.items()}})
        _lib.check(self.L.ldpc_ctx_set_pb_tuning(self._ctx, C.byref(t)), "ldpc_ctx_set_pb_tuning")
        return prev

    def osd_index_errors(self):
        """Out-of-range frame-list entries met by calls that carried ``y_frames`` (synchronises the device)."""
        n = C.c_int64(0)
        _lib.check(self.L.ldpc_osd_index_errors(self._ctx, C.byref(n)), "ldpc_osd_index_errors")
        return int(n.value)

    def osd_ge(self, rows):
        """Device GF(2) elimination of [F,64,2] packed matrices -> (reduced, swaps[F,64,2] u8, nswaps[F])."""
        self._chk(rows, torch.int64, (64, 2), "rows")
        F = rows.shape[0]
        red = self.empty((F, 64, 2), torch.int64)
        swaps = torch.zeros((F, 64, 2), dtype=torch.uint8, device=self.device)
        ns = self.empty((F,), torch.int32)
        _lib.check(self.L.ldpc_osd_ge(self._ctx, _ptr(rows), F, _ptr(red), _ptr(swaps), _ptr(ns), self._stream()),
                   "ldpc_osd_ge")
        return red, swaps, ns

    def osd_front(self, y, index=None, count=None, F=None, out=None):
        """Reliability sort + elimination + MRB bookkeeping.  Returns (perm[F,128] u8,
        parity[F,64] int64 rows of P', nswaps[F] int32); ``out`` may carry those three preallocated."""
        self._chk(y, torch.float32, (self.n,), "y")
        F = (index.shape[0] if index is not None else y.shape[0]) if F is None else F
        if out is not None:
            perm, parity, ns = out
        else:
            perm = self.empty((F, 128), torch.uint8)
            parity = self.empty((F, 64), torch.int64)
            ns = self.empty((F,), torch.int32)
        _lib.check(self.L.ldpc_osd_front(self._ctx, _ptr(y), _ptr(index), _ptr(count), F, _ptr(perm), _ptr(parity),
                                         _ptr(ns), self._stream()), "ldpc_osd_front")
        return perm, parity, ns

    def osd_params(self, order, algo=_lib.OSD_CONVENTIONAL, snr_db=0.0, fs_beta=0.1, fs_tau_e=6.5, fs_tau_psc=30.0,
                   fs_reference_quirk=1, aux=None, table_scan=False, pb_path=None, readlane_scan=False, y_frames=0,
                   pb_front_inside=False):
        """aux: optional int32 tensor [F,4] receiving the PB-OSD per-frame statistics;
        table_scan: use the table-driven conventional kernel also for order 2 (cross-check path);
        pb_path: None = staged PB-OSD kernels, "block" = every frame through the sorted-chunk kernel from its
        first TEP, "replay" = every frame through the literal list replay (cross-check paths);
        readlane_scan: conventional order 2 through the first register-resident kernel (triangular pairing by
        v_readlane) instead of the rotation-paired persistent one (cross-check path);
        y_frames: debug bound for caller-made frame lists (0 = off): entries of ``index`` outside [0, y_frames) are
        replaced by 0 and counted (``osd_index_errors``);
        pb_front_inside: PB-OSD through ``osd_decode`` with the front end inside the first PB kernel (nothing goes through a
        workspace: 43 % less HBM traffic for front end + head, 4-5 % more time; the same bit as table_scan)."""
        flags = (1 if (table_scan or pb_front_inside) else 0) | {None: 0, "block": 2, "replay": 4}[pb_path] | (8 if readlane_scan else 0)
        return _lib.OsdParams(int(order), int(algo), float(snr_db), float(fs_beta), float(fs_tau_e),
                              float(fs_tau_psc), int(fs_reference_quirk), flags,
                              aux.data_ptr() if aux is not None else None, int(y_frames))

    def osd_decode(self, y, order, algo=_lib.OSD_CONVENTIONAL, index=None, count=None, F=None, params=None, out=None):
        """OSD of the frames y[index[f]] (or y[f]).  Returns dict(cw[F,2] int64 original bit
        order, metric[F] f32, best[F] i32, ntep[F] i32)."""
        self._chk(y, torch.float32, (self.n,), "y")
        F = (index.shape[0] if index is not None else y.shape[0]) if F is None else F
        p = params if params is not None else self.osd_params(order, algo)
        out = dict(out or {})
        if out.get("cw") is None:
            out["cw"] = self.empty((F, 2), torch.int64)
        if out.get("metric") is None:
            out["metric"] = self.empty((F,), torch.float32)
        if out.get("best") is None:
            out["best"] = self.empty((F,), torch.int32)
        if out.get("ntep") is None:
            out["ntep"] = self.empty((F,), torch.int32)
        _lib.check(self.L.ldpc_osd_decode(self._ctx, _ptr(y), _ptr(index), _ptr(count), F, C.byref(p), _ptr(out["cw"]),
                                          _ptr(out["metric"]), _ptr(out["best"]), _ptr(out["ntep"]), self._stream()),
                   "ldpc_osd_decode")
        return out

    def osd_search(self, y, perm, parity, params, index=None, count=None, F=None, out=None):
        """Search only, on caller-supplied front-end results (perm [F,128] u8, parity [F,64] int64)."""
        self._chk(y, torch.float32, (self.n,), "y")
        self._chk(perm, torch.uint8, (128,), "perm")
        self._chk(parity, torch.int64, (64,), "parity")
        F = perm.shape[0] if F is None else F
        out = dict(out or {})
        for name, shape, dt in (("cw", (F, 2), torch.int64), ("metric", (F,), torch.float32),
                                ("best", (F,), torch.int32), ("ntep", (F,), torch.int32)):
            if out.get(name) is None:
                out[name] = self.empty(shape, dt)
        _lib.check(self.L.ldpc_osd_search(self._ctx, _ptr(y), _ptr(index), _ptr(count), F, _ptr(perm), _ptr(parity),
                                          C.byref(params), _ptr(out["cw"]), _ptr(out["metric"]), _ptr(out["best"]),
                                          _ptr(out["ntep"]), self._stream()), "ldpc_osd_search")
        return out

    def osd_tep_eval(self, y, perm, parity, mask, index=None, count=None):
        """One given TEP per frame (mask [F] int64: bit p flips primed MRB position p) on front-end results.
        Returns dict(cw[F,2] int64 original bit order, metric[F] f32, hd[F] i32)."""
        self._chk(y, torch.float32, (self.n,), "y")
        self._chk(perm, torch.uint8, (128,), "perm")
        self._chk(parity, torch.int64, (64,), "parity")
        self._chk(mask, torch.int64, (), "mask")
        F = perm.shape[0]
        out = dict(cw=self.empty((F, 2), torch.int64), metric=self.empty((F,), torch.float32), hd=self.empty((F,), torch.int32))
        _lib.check(self.L.ldpc_osd_tep_eval(self._ctx, _ptr(y), _ptr(index), _ptr(count), F, _ptr(perm), _ptr(parity), _ptr(mask),
                                            _ptr(out["cw"]), _ptr(out["metric"]), _ptr(out["hd"]), self._stream()), "ldpc_osd_tep_eval")
        return out

    # ------------------------------------------------------------------ H-form OSD (DL-OSD stage)
    def hosd_front(self, order_llr):
        """check_matrix_reorder + identify_mrb on [F,128] ordering values.  Returns (lri[F,128] u8,
        uidx[F,128] u8, M[F,64] int64 rows of updated_M, nswaps[F] int32)."""
        self._chk(order_llr, torch.float32, (self.n,), "order_llr")
        F = order_llr.shape[0]
        lri = self.empty((F, 128), torch.uint8)
        uidx = self.empty((F, 128), torch.uint8)
        M = self.empty((F, 64), torch.int64)
        ns = self.empty((F,), torch.int32)
        _lib.check(self.L.ldpc_hosd_front(self._ctx, _ptr(order_llr), F, _ptr(lri), _ptr(uidx), _ptr(M), _ptr(ns),
                                          self._stream()), "ldpc_hosd_front")
        return lri, uidx, M, ns

    def hosd_search(self, order_llr, metric_llr, front, teps, block_off, label_bits=None, want_arg=True, want_best=True):
        """Block minima over the TEP blocks ``teps[block_off[b]:block_off[b+1]]`` (device tensors: [Nt,4] u8,
        [nblk+1] int32).  Returns dict(block_min[F,nblk], block_arg, truth, cw, metric, best)."""
        self._chk(order_llr, torch.float32, (self.n,), "order_llr")
        self._chk(metric_llr, torch.float32, (self.n,), "metric_llr")
        lri, uidx, M = front[:3]
        F = order_llr.shape[0]
        if metric_llr.shape[0] != F or lri.shape[0] != F:
            raise ValueError("order_llr, metric_llr and the front-end results must hold the same frames")
        self._chk(teps, torch.uint8, (4,), "teps")
        if block_off.dtype != torch.int32 or block_off.device != self.device or block_off.dim() != 1 or block_off.numel() < 1:
            raise ValueError("block_off: expected a 1-D int32 tensor [nblk+1] on the device")
        nblk = block_off.numel() - 1
        out = dict(block_min=self.empty((F, nblk), torch.float32),
                   block_arg=self.empty((F, nblk), torch.int32) if want_arg else None,
                   truth=self.empty((F,), torch.float32) if label_bits is not None else None,
                   cw=self.empty((F, 2), torch.int64) if want_best else None,
                   metric=self.empty((F,), torch.float32) if want_best else None,
                   best=self.empty((F,), torch.int32) if want_best else None)
        _lib.check(self.L.ldpc_hosd_search(self._ctx, _ptr(order_llr), _ptr(metric_llr), F, _ptr(lri), _ptr(uidx), _ptr(M),
                                           _ptr(teps), _ptr(block_off), nblk, _ptr(label_bits), _ptr(out["block_min"]),
                                           _ptr(out["block_arg"]), _ptr(out["truth"]), _ptr(out["cw"]), _ptr(out["metric"]),
                                           _ptr(out["best"]), self._stream()), "ldpc_hosd_search")
        return out

    def osd_counts(self, cw, label_bits, index=None, count=None, ntep=None, counts=None, F=None):
        """counts[3] += {frames, frames_wrong, teps_total}; labels are looked up through index."""
        F = cw.shape[0] if F is None else F
        if counts is None:
            counts = torch.zeros(3, dtype=torch.int64, device=self.device)
        _lib.check(self.L.ldpc_osd_counts(self._ctx, _ptr(cw), _ptr(label_bits), _ptr(index), _ptr(count), _ptr(ntep),
                                          F, _ptr(counts), self._stream()), "ldpc_osd_counts")
        return counts


_default = {}


def default_decoder(code: Code, device=None) -> Decoder:
    """Decoder cache keyed by (code object, device) for the reference-style global-state API."""
    dev = torch.cuda.current_device() if device is None else int(device)
    key = (id(code), dev)
    if key not in _default:
        _default[key] = Decoder(code, dev)
    return _default[key]
