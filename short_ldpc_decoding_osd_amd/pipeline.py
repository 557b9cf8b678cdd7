"""One-call batch pipeline: NMS -> counters -> failure compaction -> OSD -> counters
(``ldpc_pipeline_run``).  This is the body of the reference drivers' batch loops
(ldpc_128_testing.py:117-131, then pb_testing.py / fs_testing.py per failed frame) executed as
seven kernel launches from ONE host call, with no device-to-host traffic in between.  All buffers are
allocated once; ``run()`` only enqueues work on torch's current stream.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .runtime import Decoder


class BatchPipeline:
    def __init__(self, dec: Decoder, B: int, T: int, alpha, osd_order=None, osd_algo=_lib.OSD_CONVENTIONAL, snr_db=0.0,
                 w_in=1.0, w_out=1.0, want_soft=True, keep_front=True, **osd_kw):
        self.dec, self.B, self.T = dec, int(B), int(T)
        e = dec.empty
        self.soft = e((B, dec.n), torch.float32) if want_soft else None
        self.hard = e((B, dec.words), torch.int64)
        self.fail = e((B,), torch.uint8)
        self._counts = torch.zeros(8, dtype=torch.int64, device=dec.device)   # one buffer: no torch op to gather them
        self.nms_counts, self.osd_counts = self._counts[:5], self._counts[5:]
        self._alpha = np.ascontiguousarray(np.broadcast_to(np.asarray(alpha, dtype=np.float32), (max(T, 1),)))
        p = _lib.Pipeline()
        p.B, p.T, p.nms_kernel = self.B, self.T, _lib.NMS_AUTO
        p.alpha = self._alpha.ctypes.data_as(C.POINTER(C.c_float))
        p.w_in, p.w_out = float(w_in), float(w_out)
        p.d_soft = self.soft.data_ptr() if want_soft else None
        p.d_hard, p.d_fail = self.hard.data_ptr(), self.fail.data_ptr()
        p.d_nms_counts = self.nms_counts.data_ptr()
        p.osd_enable, p.timing_slot = int(osd_order is not None), -1
        self.osd_order = osd_order
        if osd_order is not None:
            self.index, self.count = e((B,), torch.int32), e((1,), torch.int32)
            # keep_front: the front-end results (perm, P' rows) land in buffers of this object and the two
            # OSD kernels are timed separately; otherwise they go through the context's workspace
            self.perm = e((B, 128), torch.uint8) if keep_front else None
            self.parity = e((B, 64), torch.int64) if keep_front else None
            self.cw, self.metric = e((B, 2), torch.int64), e((B,), torch.float32)
            self.best, self.ntep = e((B,), torch.int32), e((B,), torch.int32)
            self.aux = torch.zeros((B, 4), dtype=torch.int32, device=dec.device) if osd_algo == _lib.OSD_PB else None
            p.osd = dec.osd_params(osd_order, osd_algo, snr_db=snr_db, aux=self.aux, **osd_kw)
            p.d_index, p.d_count = self.index.data_ptr(), self.count.data_ptr()
            p.d_perm = self.perm.data_ptr() if keep_front else None
            p.d_parity = self.parity.data_ptr() if keep_front else None
            p.d_cw, p.d_metric = self.cw.data_ptr(), self.metric.data_ptr()
            p.d_best, p.d_ntep = self.best.data_ptr(), self.ntep.data_ptr()
            p.d_osd_counts = self.osd_counts.data_ptr()
        self._p = p
        self._y = self._labels = None

    def bind(self, y, label_bits=None):
        """Attach the input batch ([B,n] f32) and optional packed labels ([B,words] int64)."""
        self.dec._chk(y, torch.float32, (self.dec.n,), "y")
        if y.shape[0] != self.B:
            raise ValueError(f"batch of {y.shape[0]} frames bound to a pipeline for {self.B}")
        self._y, self._labels = y, label_bits
        self._p.d_llr = y.data_ptr()
        self._p.d_label_bits = label_bits.data_ptr() if label_bits is not None else None
        return self

    def run(self, timing_slot=-1):
        if self._y is None:
            raise RuntimeError("bind() a batch first")
        self._p.timing_slot = int(timing_slot)
        _lib.check(self.dec.L.ldpc_pipeline_run(self.dec._ctx, C.byref(self._p), self.dec._stream()), "ldpc_pipeline_run")

    def timing(self, slot):
        """(nms_ms, osd_front_ms, osd_search_ms) of the run that used ``slot``; synchronise first."""
        ms = (C.c_float * 3)()
        _lib.check(self.dec.L.ldpc_pipeline_timing(self.dec._ctx, int(slot), ms), "ldpc_pipeline_timing")
        return tuple(float(v) for v in ms)

    def reset_counters(self):
        self._counts.zero_()

    def counters(self):
        """int64[8]: {frames, frame_err, bit_err, undetected, synd_fail, osd_frames, osd_wrong, teps}."""
        return self._counts
