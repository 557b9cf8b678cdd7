"""NMS test-stage decoder with the reference's call surface
(LDPC_128/Ldpc_128_testing/ms_test.py), running on libldpcosd.so / MI355X.

    GL.set_map('code_parameters', Code(...)); GL.set_map('num_iterations', 10)
    GL.set_map('selected_decoder_type', 'NMS-1')
    model = Decoding_model()
    fer, ber, undetected, (buffer_inputs, buffer_labels) = model(inputs, labels)

Same names, arity and return shapes as the reference; tensors come back as NumPy arrays, the two row buffers of
``Decoding_model.call`` as ``RowBuffer`` sequences (list-like views of one array each).
The learned weights are plain attributes holding the *stored* (pre-softplus) values, as in
the TF checkpoint (ms_test.py:83, :207-208): ``layer.shared_check_weight`` etc.
"""
from __future__ import annotations

import collections.abc

import numpy as np
import torch

from . import globalmap as GL
from .runtime import default_decoder
from .weights import softplus32 as _softplus32


class RowBuffer(collections.abc.Sequence):
    """The list of rows ``Decoding_model.call`` returns (buffer_inputs / buffer_labels, ms_test.py:55-64), held as ONE 2-D array
    instead of a Python list of row objects: it has ``len``, indexing, slicing and iteration like the reference's lists, so
    ``np.stack(buf)``, ``buf[k]``, ``for row in buf`` and the reference's flattening comprehension all work -- but building
    370 k row objects per 131 072-frame batch (and a 380 MB int64 copy of the labels repeated T + 1 times) was 60 % of the
    call.  ``repeat`` > 1: row k is ``array[k // repeat]`` (the label of a failed frame stands for its T + 1 rows).
    ``materialize()`` returns the plain 2-D array (with the repeats written out)."""

    def __init__(self, array, repeat=1):
        self.array, self.repeat = array, int(repeat)

    def __len__(self):
        return self.array.shape[0] * self.repeat

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[i] for i in range(*k.indices(len(self)))]
        if k < 0:
            k += len(self)
        if not 0 <= k < len(self):
            raise IndexError(k)
        return self.array[k // self.repeat]

    def __iter__(self):
        if self.repeat == 1:
            return iter(self.array)
        return (self.array[k // self.repeat] for k in range(len(self)))

    def materialize(self):
        return self.array if self.repeat == 1 else np.repeat(self.array, self.repeat, axis=0)

    @staticmethod
    def concatenate(parts):
        """One buffer from several (postprocess_failure_cases): arrays concatenated, repeats written out only if they differ."""
        parts = list(parts)
        if parts and all(isinstance(p, RowBuffer) for p in parts) and len({p.repeat for p in parts}) == 1:
            return RowBuffer(np.concatenate([p.array for p in parts], axis=0), parts[0].repeat)
        return [row for p in parts for row in p]


class Decoder_Layer:
    """``Decoder_Layer`` (ms_test.py:72-242).  Supports NMS-1 / NMS-2 / NMS-3 (:82-91)."""

    def __init__(self, initial_value=-0.048):
        self.decoder_type = GL.get_map('selected_decoder_type')
        self.num_iterations = GL.get_map('num_iterations')
        self.code = GL.get_map('code_parameters')
        self.feature_len = self.code.max_chk_degree - 1
        self.initials = initial_value
        self.build(None)

    def build(self, input_shape):
        init = np.full([1], self.initials, dtype=np.float32)
        t = self.decoder_type
        if t == 'NMS-1':
            self.shared_check_weight = init.copy()
        elif t == 'NMS-2':
            self.shared_bit_weight = init.copy()
            self.shared_check_weight = init.copy()
        elif t == 'NMS-3':
            self.shared_bit_weight1 = init.copy()
            self.shared_bit_weight2 = init.copy()
            self.shared_check_weight = init.copy()
        else:
            raise NotImplementedError(f"decoder type '{t}': only NMS-1/2/3 run on the HIP path "
                                      "(NMS-r / compute_cv1, ms_test.py:146-178, is not exercised by any driver)")

    # effective (post-softplus) factors, ms_test.py:127-131, :207-208, :222-225
    def effective_weights(self):
        alpha = _softplus32(self.shared_check_weight[0])
        w_in = w_out = np.float32(1.0)
        if self.decoder_type == 'NMS-2':
            w_in = w_out = _softplus32(self.shared_bit_weight[0])
        if self.decoder_type == 'NMS-3':
            w_in = _softplus32(self.shared_bit_weight1[0])
            w_out = _softplus32(self.shared_bit_weight2[0])
        return alpha, w_in, w_out

    def _device_run(self, soft_input, want_traj):
        dec = default_decoder(self.code)
        if isinstance(soft_input, torch.Tensor):
            y = soft_input.to(device=dec.device, dtype=torch.float32).contiguous()
        else:
            y = torch.from_numpy(np.ascontiguousarray(soft_input, dtype=np.float32)).to(dec.device)
        alpha, w_in, w_out = self.effective_weights()
        res = dec.nms(y, self.num_iterations, alpha, w_in, w_out, want_soft=True, want_traj=want_traj)
        return dec, y, res

    def call(self, soft_input, labels=None):
        """-> [soft_input, out_1, ..., out_T]  (ms_test.py:99-121, bp_result[4])"""
        dec, y, res = self._device_run(soft_input, want_traj=True)
        traj = _to_host(res["traj"])
        return [_to_host(y)] + [traj[i] for i in range(self.num_iterations)]

    __call__ = call


def _to_host(t):
    """Device tensor -> NumPy array; large ones through a page-locked block (a pageable copy allocates and faults its target in
    page by page on every call: 25 ms per 190 MB, against 3.4 ms -- scripts/profile_surface.py)."""
    if t.numel() * t.element_size() < (1 << 21):
        return t.cpu().numpy()
    host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    host.copy_(t)
    return host.numpy()


class Decoding_model:
    """``Decoding_model`` (ms_test.py:26-70)."""

    def __init__(self):
        self.layer = Decoder_Layer()

    def set_check_weight(self, stored_value):
        """Stored (pre-softplus) value of ``shared_check_weight`` -- what the checkpoint holds."""
        self.layer.shared_check_weight = np.full([1], stored_value, dtype=np.float32)

    def call(self, inputs, labels):
        """``(fer, ber, undetected, (buffer_inputs, buffer_labels))`` of one batch (ms_test.py:30-34).

        Device sequence: NMS-T (posterior, hard words, syndrome flags -- NO trajectory) -> error counters -> failed-frame
        list -> the T + 1 rows of the FAILED frames only (``ldpc_nms_traj_rows``: the listed frames decoded again, rows in
        the reference's buffer order) -> one device-to-host copy of [F, T+1, n].  Round 3 wrote the [T][B][n] trajectory
        of every frame (5 KiB per input frame) and gathered the failures afterwards; the reference keeps T + 1 rows of the
        failed frames only (collect_failed_output_selective, :55-64): 1.4 KiB per input frame at 2.5 dB."""
        layer = self.layer
        dec, y, res = layer._device_run(inputs, want_traj=False)
        if isinstance(labels, torch.Tensor):
            lab_t = labels.to(device=dec.device, dtype=torch.int64).contiguous()
            labels_np = None
        else:
            labels_np = np.ascontiguousarray(labels, dtype=np.int64)
            lab_t = torch.from_numpy(labels_np).to(dec.device)
        label_bits = dec.pack_bits(lab_t)
        counts_d = dec.eval_counts(res["hard"], label_bits, res["fail"])
        index, count = dec.compact(res["fail"])
        B, n = y.shape
        T = layer.num_iterations
        alpha, w_in, w_out = layer.effective_weights()
        host = torch.cat([counts_d, count.to(torch.int64)]).cpu().numpy()      # ONE small copy: the five counters and the failure count
        counts, nfail = host[:5], int(host[5])
        rows_np = np.zeros((0, n), np.float32)
        idx_np = np.zeros((0,), np.int64)
        failed_labels_np = np.zeros((0, n), np.int64)
        if nfail:
            rows = dec.nms_traj_rows(y, index, count, nfail, T, alpha, w_in, w_out)      # [F, T+1, n]
            # (into page-locked memory: the copy is the largest item of a call -- 190 MB for 131 072 frames at 2.5 dB -- and runs at
            #  twice the rate of a pageable one; torch's host allocator recycles the block once the caller drops the array)
            big = nfail * (T + 1) * n * 4 >= (1 << 21)      # (the reference's batch of 1000 stays on the plain path: 0.33 ms a call)
            rows_host = torch.empty((nfail * (T + 1), n), dtype=torch.float32, pin_memory=big)
            rows_host.copy_(rows.reshape(nfail * (T + 1), n))
            rows_np = rows_host.numpy()
            idx_np = index[:nfail].cpu().numpy().astype(np.int64)
            # the failed frames' label rows: gathered on the device (they are there already) and copied the same way -- NumPy's
            # fancy index over 134 MB of int64 rows took 3 ms of a 13 ms call
            if big:
                lab_host = torch.empty((nfail, n), dtype=torch.int64, pin_memory=True)
                lab_host.copy_(lab_t.index_select(0, index[:nfail].to(torch.int64)))
                failed_labels_np = lab_host.numpy()
            else:
                failed_labels_np = (labels_np if labels_np is not None else lab_t.cpu().numpy())[idx_np]
        fer = float(counts[1]) / B                       # 1 - len(success_index)/B       (:52)
        ber = float(counts[2]) / (B * n)                 # (:53)
        undetected = int(counts[3])                      # len(not_in_success_index)      (:46-54)
        buffer_inputs = RowBuffer(rows_np)               # T+1 rows per failed frame, row 0 = channel (:55-64)
        buffer_labels = RowBuffer(failed_labels_np, repeat=T + 1)
        self.last_counts = dict(zip(("frames", "frame_err", "bit_err", "undetected", "synd_fail"),
                                    (int(c) for c in counts)))
        self.last_failed_index = idx_np
        return fer, ber, undetected, (buffer_inputs, buffer_labels)

    __call__ = call

    def get_eval(self, soft_output_list, labels):
        """(FER, BER, undetected, index[F,1]) from a soft-output list (ms_test.py:36-54)."""
        code = GL.get_map('code_parameters')
        dec = default_decoder(code)
        soft = torch.from_numpy(np.ascontiguousarray(soft_output_list[-1], dtype=np.float32)).to(dec.device)
        res = dec.nms(soft, 0, 1.0, want_soft=False)      # T = 0: hard decision + syndrome only
        lab = torch.from_numpy(np.ascontiguousarray(labels, dtype=np.int64)).to(dec.device)
        counts = dec.eval_counts(res["hard"], dec.pack_bits(lab), res["fail"]).cpu().numpy()
        index, count = dec.compact(res["fail"])
        idx = index[: int(count.cpu()[0])].cpu().numpy().astype(np.int64)
        B, n = soft.shape
        return float(counts[1]) / B, float(counts[2]) / (B * n), int(counts[3]), idx.reshape(-1, 1)

    def collect_failed_output_selective(self, soft_output_list, labels, index):
        list_length = self.layer.num_iterations + 1
        buffer_inputs, buffer_labels = [], []
        for i in np.asarray(index).reshape(-1):
            for j in range(list_length):
                buffer_inputs.append(soft_output_list[j][i])
                buffer_labels.append(labels[i])
        return buffer_inputs, buffer_labels

    def postprocess_failure_cases(self, buffer):
        """Flatten the per-batch lists (ms_test.py:66-70)."""
        return RowBuffer.concatenate(buffer[0]), RowBuffer.concatenate(buffer[1])


def calculation_loss(soft_output, labels):
    """Summed sigmoid cross entropy with logits = -soft_output (ms_test.py:245-249)."""
    x = -np.asarray(soft_output, dtype=np.float64)
    z = np.asarray(labels, dtype=np.float64)
    return np.float32(np.sum(np.maximum(x, 0) - x * z + np.log1p(np.exp(-np.abs(x)))))


def save_decoded_data(updated_buffer, file_dir, snr, log_filename, list_length):
    """Per-iteration mean CE to the log, then the retest TFRecord (ms_test.py:251-272)."""
    from . import data_generating as Data_gen

    def as_array(buf, dtype):
        if isinstance(buf, RowBuffer):
            return buf.materialize()
        return np.stack(buf) if len(buf) else np.zeros((0, 0), dtype)

    info, label = as_array(updated_buffer[0], np.float32), as_array(updated_buffer[1], np.int64)
    CE_loss_list = []
    tested = 0
    for i in range(list_length):
        bits, labs = info[i::list_length], label[i::list_length]
        tested = bits.shape[0]
        CE_loss_list.append(calculation_loss(bits, labs) / max(tested, 1))
    print(CE_loss_list)
    with open(log_filename, 'a+') as f:
        f.write(str(tested) + 'tested:\n')
        f.write("# CE list:\n")
        f.write(' '.join(map(str, CE_loss_list)) + '\n')
    print("%.4f tested\nCE_list:%s" % (tested, str(CE_loss_list)))
    print("Data for retraining  with %d cases to be stored " % info.shape[0])
    Data_gen.make_tfrecord((info, label), out_filename=file_dir)
    print('For ' + str(round(snr, 2)) + "dB:Data storing finished!")
