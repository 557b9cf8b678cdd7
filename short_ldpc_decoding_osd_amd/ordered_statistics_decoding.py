"""Mirror of the DL-OSD stage's H-form OSD class:
LDPC_128/DL_OSD_Testing_serial/ordered_statistics_decoding.py (class ``osd``) plus the pattern helpers
of nn_testing.py:65-82,120-148 and globalmap.py:57-76 that feed it.

The stage's trained networks (the bit-wise CNN that refines the LLRs and the sliding-window
classifier ``fcn``) stay ordinary Python callables on the host, as in the reference; everything
per frame and per TEP -- ascending reliability sort, elimination of the permuted H, MRB bookkeeping,
re-encoding and weighted-distance scan of every TEP block -- runs in the HIP kernels behind
``ldpc_hosd_front`` / ``ldpc_hosd_search``.  ``sliding_osd`` evaluates ALL blocks of the decoding
path on the device (the reference evaluates them lazily) and then replays the reference's window
loop on the block minima, so decisions, window counts and complexity figures are the reference's.

Only the live path of the reference is mirrored (``Testing_OSD`` -> ``sliding_osd``, nn_testing.py:214);
``execute_osd`` / ``best_estimating`` / ``collect_tep`` are never called there.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from . import globalmap as GL
from ._osd_common import _dec, _unpack_rows, full_gf2elim as _full_gf2elim


def secure_segment_threshold():
    """DL_OSD_Testing_serial/globalmap.py:57-76 -> (segment sizes, MRB boundaries); reads
    'segment_num' and 'code_parameters' from the settings map."""
    num_seg = GL.get_map('segment_num')
    code = GL.get_map('code_parameters')
    allocation_length = code.k - 1
    basic_length = list(range(1, num_seg))
    num_basic = sum(basic_length)
    sizes = [int(allocation_length / num_basic * b) for b in basic_length]
    sizes[-1] += allocation_length - sum(sizes)
    whole = np.insert(np.array(sizes, dtype=np.int64), 0, 1)
    return whole, np.insert(np.cumsum(whole), 0, 0)


def query_convention_path(order_sum=None):
    """nn_testing.py:65-82 (the path itself; the reference also returns the network's name):
    3-segment order patterns of total weight <= i for i = 0..threshold_sum, first occurrence kept."""
    order_sum = (GL.get_map('threshold_sum') if order_sum is None else order_sum) + 1
    path, seen = [], set()
    for i in range(order_sum):
        for j1 in range(order_sum):
            for j2 in range(order_sum):
                for j3 in range(order_sum):
                    if j1 + j2 + j3 <= i and (j1, j2, j3) not in seen:
                        seen.add((j1, j2, j3))
                        path.append([j1, j2, j3])
    return path


def filter_order_patterns(decoding_path):
    """nn_testing.py:110-118: keep patterns of weight <= 'threshold_sum', first 'decoding_length' of them."""
    threshold_sum = GL.get_map('threshold_sum')
    return [p for p in decoding_path if sum(p) <= threshold_sum][:GL.get_map('decoding_length')]


def generate_teps(osd_instance, residual_path):
    """nn_testing.py:138-148: one TEP block per order pattern of the path -> (list of [N_b,k] arrays,
    cumulative block sizes with a leading 0)."""
    num_segments = GL.get_map('segment_num')
    _, boundary = secure_segment_threshold()
    range_list = [range(int(boundary[i]), int(boundary[i + 1])) for i in range(num_segments)]
    blocks = [osd_instance.error_pattern_gen(p, range_list) for p in residual_path]
    return blocks, np.insert(np.cumsum([b.shape[0] for b in blocks]), 0, 0)


def _teps_from_matrix(E):
    """[N,k] 0/1 pattern matrix -> [N,4] u8 {p0,p1,p2,weight} for the device scan."""
    E = np.asarray(E)
    wt = E.sum(axis=1)
    if E.shape[0] and wt.max() > 3:
        raise _lib.LdpcError("TEP blocks of weight > 3 are not supported by the device scan")
    out = np.zeros((E.shape[0], 4), dtype=np.uint8)
    rows, cols = np.nonzero(E)            # row-major: positions ascend inside a row
    first = np.searchsorted(rows, np.arange(E.shape[0]))
    out[rows, np.arange(rows.size) - first[rows]] = cols
    out[:, 3] = wt
    return out


class osd:   # noqa: N801  (name kept from the reference)
    def __init__(self, code):
        self.original_H = code.H
        self.n_dims = code.check_matrix_column
        self.k = code.k
        self.m = self.n_dims - self.k
        self.last = None
        self._blocks_key = None

    # ---------------------------------------------------------------- per-frame pieces
    def mag_input_gen(self, inputs):
        """:25-28 -- ascending |inputs| argsort (ties: lower index first), computed on the device."""
        dec = _dec()
        x = torch.from_numpy(np.ascontiguousarray(np.atleast_2d(np.asarray(inputs, dtype=np.float32)))).to(dec.device)
        return dec.hosd_front(x)[0].cpu().numpy().astype(np.int64)

    def check_matrix_reorder(self, iteration_inputs, inputs, labels):
        """:30-41 -> (order_H_list [F,m,n], order_inputs, order_original_list, order_labels)."""
        inputs = np.asarray(inputs, dtype=np.float32)
        list_length = GL.get_map('num_iterations') + 1
        lri_p = self.mag_input_gen(inputs)
        take = lambda a: np.take_along_axis(np.asarray(a), lri_p, axis=1)   # noqa: E731
        order_original_list = [take(np.asarray(iteration_inputs)[i::list_length]) for i in range(list_length)]
        order_H_list = np.asarray(self.original_H)[:, lri_p].transpose(1, 0, 2)
        return order_H_list, take(inputs), order_original_list, take(labels)

    def full_gf2elim(self, M):
        """:222-257 (the elimination rule shared by every stage; device for 64 x 128)."""
        return _full_gf2elim(M)

    def identify_mrb(self, order_H_list):
        """:43-80 -> (updated_index_list, updated_M_list, swap_len_list, swap_lrb_position_list)."""
        code = GL.get_map('code_parameters')
        n, k = code.check_matrix_column, code.k
        threshold_sum = GL.get_map('threshold_sum')
        idx_list, M_list, len_list, pos_list = [], [], [], []
        for H_i in np.asarray(order_H_list):
            R, swaps = self.full_gf2elim(np.array(H_i, dtype=np.int64))
            index_order = np.arange(n)
            for a, b in swaps:
                index_order[a], index_order[b] = index_order[b], index_order[a]
            updated_MRB, updated_LRB = index_order[-k:], index_order[:k]
            sw = np.argsort(updated_MRB, kind="stable")
            idx_list.append(np.concatenate([index_order[:n - k], updated_MRB[sw]]))
            M_list.append(R[:, -k:][:, sw])
            len_list.append(int(np.where(updated_MRB >= n - k, 0, 1).sum()))
            pos_list.append(np.where(updated_LRB >= (n - k) - 4 * threshold_sum, 1, 0))
        return idx_list, M_list, len_list, pos_list

    def error_pattern_gen(self, direction, range_list):
        """:81-98 -> int array [N,k]; the enumeration is the library's (``ldpc_hosd_pattern_teps``).
        ``range_list`` must be consecutive ranges (as every caller in the reference builds them)."""
        bounds = [range_list[0].start] + [r.stop for r in range_list]
        if any(range_list[i].start != bounds[i] or range_list[i].step != 1 for i in range(len(range_list))):
            raise ValueError("error_pattern_gen: segments must be consecutive unit-step ranges")
        L = _lib.load()
        nseg = len(range_list)
        b = (C.c_int32 * (nseg + 1))(*bounds)
        p = (C.c_int32 * nseg)(*[int(v) for v in direction])
        n = _lib.check(L.ldpc_hosd_pattern_teps(nseg, b, p, None), "ldpc_hosd_pattern_teps")
        t = np.zeros((max(n, 1), 4), dtype=np.uint8)
        _lib.check(L.ldpc_hosd_pattern_teps(nseg, b, p, t.ctypes.data_as(C.POINTER(C.c_uint8))), "ldpc_hosd_pattern_teps")
        E = np.zeros((n, self.k), dtype=int)
        for q in range(3):
            sel = t[:n, 3] > q
            E[np.nonzero(sel)[0], t[:n][sel, q]] = 1
        return E

    def acquire_min(self, error_pattern_matrix, initial_mrb, M_matrix, order_hard_original, mag_metric):
        """:153-162 for one frame in the updated order: min over the block of the weighted distance of
        [M.(e^mrb0), e^mrb0] to ``order_hard_original``.  One device call; the signs of two synthetic
        value vectors carry the two hard-decision vectors."""
        dec = _dec()
        mag = np.asarray(mag_metric, dtype=np.float32)
        hard = np.asarray(order_hard_original).astype(bool)
        metric = np.where(hard, -mag, mag).astype(np.float32)       # (a zero weight costs nothing under either sign)
        order = np.ones(self.n_dims, dtype=np.float32)
        order[self.m:] = np.where(np.asarray(initial_mrb).astype(bool), -1.0, 1.0)
        ident = torch.arange(128, dtype=torch.uint8, device=dec.device)[None].contiguous()
        Mrows = np.packbits(np.ascontiguousarray(M_matrix, dtype=np.uint8), axis=1, bitorder="little").view(np.int64).reshape(1, 64)
        teps = torch.from_numpy(_teps_from_matrix(error_pattern_matrix)).to(dec.device)
        off = torch.tensor([0, teps.shape[0]], dtype=torch.int32, device=dec.device)
        out = dec.hosd_search(torch.from_numpy(order[None]).to(dec.device), torch.from_numpy(metric[None]).to(dec.device),
                              (ident, ident, torch.from_numpy(Mrows).to(dec.device)), teps, off, want_arg=False,
                              want_best=False)
        return np.float32(out["block_min"].cpu().numpy()[0, 0])

    def sliding_window_ops(self, fcn, window, global_min, k):
        """:140-151 -- the early-termination network on the sorted window plus the window index."""
        window = np.asarray(window, dtype=np.float32)
        x = np.append(np.sort(window), np.float32(k)).reshape(1, -1)
        output_prb = np.asarray(fcn(x)).reshape(-1)
        early_termination = bool(output_prb[1] > GL.get_map('soft_margin'))
        return early_termination, min(global_min, window.min())

    # ---------------------------------------------------------------- the batch path
    def _device_blocks(self, dec, teps_list):
        key = (id(teps_list), len(teps_list), dec.device)
        if self._blocks_key != key:
            tabs = [_teps_from_matrix(E) for E in teps_list]
            off = np.insert(np.cumsum([t.shape[0] for t in tabs]), 0, 0).astype(np.int32)
            cat = np.concatenate(tabs) if tabs else np.zeros((0, 4), np.uint8)
            self._blocks = (torch.from_numpy(np.ascontiguousarray(cat)).to(dec.device), torch.from_numpy(off).to(dec.device))
            self._blocks_key = key
        return self._blocks

    def block_minima(self, input_list, inputs, labels, teps_list):
        """Device part of ``sliding_osd`` (:168-186 and every ``acquire_min``): -> dict of NumPy arrays
        block_min [F,L], block_arg, truth [F], cw [F,2] (packed, original bit order), metric, best,
        lri, uidx, nswaps."""
        dec = _dec()
        list_length = GL.get_map('num_iterations') + 1
        order = torch.from_numpy(np.ascontiguousarray(inputs, dtype=np.float32)).to(dec.device)
        metric = torch.from_numpy(np.ascontiguousarray(np.asarray(input_list, dtype=np.float32)[0::list_length])).to(dec.device)
        lab = dec.pack_bits(torch.from_numpy(np.ascontiguousarray(labels).astype(np.uint8)).to(dec.device))
        teps, off = self._device_blocks(dec, teps_list)
        front = dec.hosd_front(order)
        out = dec.hosd_search(order, metric, front, teps, off, label_bits=lab)
        res = {k: v.cpu().numpy() for k, v in out.items() if v is not None}
        res.update(lri=front[0].cpu().numpy(), uidx=front[1].cpu().numpy(), nswaps=front[3].cpu().numpy())
        return res

    def sliding_osd(self, fcn, input_list, inputs, labels, tep_info):
        """:164-220 -> (success_dec, failure_dec, windows_sum, complexity_sum); ``self.last`` keeps the
        per-frame arrays (block minima, truth metric, best codeword ...) the reference discards."""
        win = GL.get_map('sliding_win_width')
        teps_list, acc_block_size = tep_info
        res = self.block_minima(input_list, inputs, labels, teps_list)
        success_dec = failure_dec = windows_sum = complexity_sum = 0
        decided = np.zeros(len(res["truth"]), dtype=bool)
        for i in range(len(res["truth"])):
            mins = res["block_min"][i]
            window = list(mins[:win])                                 # :188-191
            global_min = min(window)
            deep_limit = win
            for k in range(len(teps_list) - win + 1):                 # :193-209
                deep_limit = k + win
                if k != 0:
                    min_sum = mins[win + k - 1]
                    window.append(min_sum)
                    window = window[-win:]
                    if min_sum > global_min:
                        continue
                early_termination, global_min = self.sliding_window_ops(fcn, window, global_min, k)
                if early_termination:
                    break
            complexity_sum += int(acc_block_size[deep_limit])         # :211-212
            windows_sum += deep_limit - win + 1
            decided[i] = bool(global_min == res["truth"][i])          # :213
            success_dec += int(decided[i])
            failure_dec += int(not decided[i])
        res["success"] = decided
        self.last = res
        return success_dec, failure_dec, windows_sum, complexity_sum

    def unpack_codewords(self, cw):
        """[F,2] packed words -> [F,128] 0/1 (original bit order)."""
        return _unpack_rows(np.ascontiguousarray(cw)[:, None, :]).reshape(len(cw), -1)[:, :self.n_dims].astype(np.int64)
