"""FS-OSD test stage with the reference's surface (LDPC_128/FS_OSD/fs_testing.py).

``fs_osd(snr, beta, selected_ds)`` -- same dataset object, ``globalmap`` switches ('fs_osd',
'convention_osd', 'miracle_view', 'order_limit', 'd_min', 'tau_psc', 'termination_num_threshlod') and
log lines (:195-231) as the reference, with whole chunks of frames decoded per device call.  The
reported S/F follow the reference's own failure test, i.e. they are computed on ``optimal_codeword``,
which a tau_e hit does not update (:143-146, :162 -- SURVEY A.4); pass ``intended=True`` to judge the
tau_e winner instead.  A summary dict is returned (the reference returns None).
"""
from __future__ import annotations

import ctypes as C
import math
import os
import time
from collections import Counter

import numpy as np

from . import _lib
from . import convention_osd as cnv_OSD
from . import globalmap as GL
from ._osd_common import (batch_osd, collect_first_rows, full_gf2elim, identify_mrb, miracle_view,  # noqa: F401
                          primed_search, swapped_info)
from .pb_testing import CHUNK, _cut, _mrb_errors


def acquire_pnc_boundary(descending_input):
    """Running float32 sums of the (i+1) least reliable MRB values (fs_testing.py:22-30)."""
    order_limit = GL.get_map('order_limit')
    k = GL.get_map('code_parameters').k
    a = np.abs(np.asarray(descending_input, dtype=np.float32))
    out = []
    for i in range(order_limit):
        acc = np.float32(0)
        for t in range(k - (i + 1), k):
            acc = np.float32(acc + a[t])
        out.append(acc)
    return out


def generate_sequential_teps(max_value, max_loops):
    """List (per weight 1..max_loops) of int32 TEP matrices in the FS visit order (:32-49)."""
    L = _lib.load()
    out = []
    for w in range(1, max_loops + 1):
        n = _lib.check(L.ldpc_tep_table_fs(max_value, w, None), "ldpc_tep_table_fs")
        sup = np.empty((n, 3), dtype=np.uint8)
        L.ldpc_tep_table_fs(max_value, w, sup.ctypes.data_as(C.POINTER(C.c_uint8)))
        m = np.zeros((n, max_value), dtype=np.int32)
        for q in range(w):
            m[np.arange(n), sup[:, q]] = 1
        out.append(m)
    return out


def one_tep_compare(updated_inputs, nth_tep, reduced_G, threshold):
    """(early_stopping, codeword [1,128], weighted distance) of one TEP (:51-64), evaluated on the device by
    ``ldpc_osd_tep_eval`` -- the searches' own LUT evaluation and float order -- on the primed-order inputs
    (identity permutation) and the parity part of ``reduced_G``.  (Not on the batched path, which lives in ``fs_osd``.)"""
    import torch
    from .runtime import default_decoder
    dec = default_decoder(GL.get_map('code_parameters'))
    y = torch.from_numpy(np.ascontiguousarray(np.asarray(updated_inputs, dtype=np.float32).reshape(1, -1))).to(dec.device)
    G = np.asarray(reduced_G, dtype=np.int64)
    k = G.shape[0]
    rows = np.packbits(G[:, k:].astype(np.uint8), axis=1, bitorder="little").view(np.uint64).reshape(1, k)
    perm = torch.arange(128, dtype=torch.uint8, device=dec.device).reshape(1, 128)
    parity = torch.from_numpy(rows.view(np.int64)).to(dec.device)
    mask = np.uint64(0)
    for p in np.flatnonzero(np.asarray(nth_tep)):
        mask |= np.uint64(1) << np.uint64(p)
    out = dec.osd_tep_eval(y, perm, parity, torch.from_numpy(np.array([mask]).view(np.int64)).to(dec.device))
    torch.cuda.synchronize()
    words = out["cw"].cpu().numpy().view(np.uint64)[0]
    cw = np.unpackbits(words.view(np.uint8), bitorder="little").astype(np.int32).reshape(1, -1)
    return bool(float(out["hd"].cpu()[0]) < threshold), cw, np.float32(out["metric"].cpu()[0])


def fs_osd(snr, beta, selected_ds, intended=False):
    start_time = time.process_time()
    order_limit = GL.get_map('order_limit')
    threshold = GL.get_map('termination_num_threshlod', 100)
    tau_psc = GL.get_map('tau_psc', 30)
    tau_e = math.floor(GL.get_map('d_min', 14) - 1) / 2          # 6.5 for d_min = 14, as the reference evaluates it (:92)
    ys, labs = collect_first_rows(selected_ds)
    summary = {}
    logdir = './log/'

    if GL.get_map('miracle_view', False):
        counter_stat = Counter(int(v) for v in _mrb_errors(ys, labs)) if len(ys) else Counter()
        print('\nFor miracle view %.1fdB (order_limit:%d) ' % (snr, order_limit) + ':')
        total_sum = sum(counter_stat.values())
        print(f'total_sum:{total_sum}')
        acc = 0
        for key, value in sorted(counter_stat.items()):
            acc += value
            print(f"order-{key}: Accumulated Ratio: {acc / total_sum:.4f}")
        summary['miracle_view'] = dict(counter_stat)
        return summary

    if GL.get_map('convention_osd', False):
        boundaries = cnv_OSD.query_boundary(order_limit)
        res = batch_osd(ys, labs, order_limit, _lib.OSD_CONVENTIONAL)
        phase = np.where(res['correct'], np.searchsorted(boundaries, res['best'], side='right'), -1)
        convention_counter = Counter(int(v) for v in phase)
        ok, bad = int(res['correct'].sum()), int((~res['correct']).sum())
        total = max(ok + bad, 1)
        FER = round(bad / total, 4)
        teps_size = boundaries[-1]
        T2 = time.process_time()
        print('\nFor Conv-OSD %.1fdB (order_limit:%d) ' % (snr, order_limit) + ':\n')
        print('----> S:' + str(ok) + ' F:' + str(bad) + '\n')
        print('Distribution of phases:' + str(convention_counter) + '\n')
        print('FER:' + str(FER) + ' Average TEPs size:', teps_size, '\n')
        os.makedirs(logdir, exist_ok=True)
        with open(logdir + 'CNV-OSD-order-' + str(order_limit) + '.txt', 'a+') as f:
            f.write('\nFor CNV-OSD %.1fdB (order_limit:%d) summary:\n' % (snr, order_limit))
            f.write('----> S:' + str(ok) + ' F:' + str(bad) + '\n')
            f.write('Distribution of phases:' + str(convention_counter) + '\n')
            f.write('FER:' + str(FER) + ' Average TEPs size:' + str(teps_size) + '\n')
            f.write(f'Running time:{T2 - start_time} seconds with mean time {(T2 - start_time) / total:.4f}!')
        summary['convention_osd'] = dict(S=ok, F=bad, FER=FER, teps=teps_size, phases=dict(convention_counter))
        return summary                                                     # the reference `continue`s (:127)

    if GL.get_map('fs_osd', False):
        fails, nteps = [], []
        for s in range(0, len(ys), CHUNK):
            r = batch_osd(ys[s:s + CHUNK], labs[s:s + CHUNK], order_limit, _lib.OSD_FS, fs_beta=float(beta),
                          fs_tau_e=float(tau_e), fs_tau_psc=float(tau_psc), fs_reference_quirk=0 if intended else 1)
            fails.append(~r['correct']); nteps.append(r['ntep'])
            if np.concatenate(fails).sum() >= threshold:
                break
        fails = np.concatenate(fails) if fails else np.zeros(0, bool)
        nteps = np.concatenate(nteps) if nteps else np.zeros(0, np.int32)
        n = _cut(fails, threshold)
        fail_sum, correct_sum = int(fails[:n].sum()), int(n - fails[:n].sum())
        total_num = max(correct_sum + fail_sum, 1)
        FER = round(fail_sum / total_num, 4)
        average_size = round(int(nteps[:n].sum()) / total_num, 5)
        T2 = time.process_time()
        print('\nFor FS-OSD %.1fdB (order_limit:%d) ' % (snr, order_limit) + ':\n')
        print('----> S:' + str(correct_sum) + ' F:' + str(fail_sum) + '\n')
        print(f'FER:{FER:.2f} Average TEPs:{average_size:.2f} \n')
        print(f'Running time:{T2 - start_time} seconds with mean time {(T2 - start_time) / total_num:.4f}!')
        os.makedirs(logdir, exist_ok=True)
        with open(logdir + 'FS-OSD-order-' + str(order_limit) + '.txt', 'a+') as f:
            f.write('\nFor FS-OSD %.1fdB (order_limit:%d) summary:\n' % (snr, order_limit))
            f.write('----> S:' + str(correct_sum) + ' F:' + str(fail_sum) + '\n')
            f.write(f'FER:{FER:.2f} Average TEPs:{average_size:.2f}\n')
            f.write(f'Running time:{T2 - start_time} seconds with mean time {(T2 - start_time) / total_num:.4f}!\n')
        summary['fs_osd'] = dict(S=correct_sum, F=fail_sum, FER=FER, average_teps=average_size, frames=n)
    return summary
