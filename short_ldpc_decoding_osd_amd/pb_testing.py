"""PB-OSD test stage with the reference's surface (LDPC_128/PB_OSD/pb_testing.py).

``pb_osd(snr, selected_ds)`` consumes the same dataset object (``as_numpy_iterator()`` yielding
(T+1)-row batches, row 0 = channel values, :71-72), honours the same ``globalmap`` switches
('pb_osd', 'convention_osd', 'miracle_view', 'order_limit', 'termination_num_threshlod') and appends
the same log lines (:190-229) -- but decodes whole chunks of frames per device call instead of one
frame per Python iteration.  The sequential stop rule "break once fail_sum >= threshold" (:174) is
applied afterwards to the ordered per-frame results, so the reported numbers are those of the
frames the reference would have visited.  A summary dict is returned (the reference returns None).
"""
from __future__ import annotations

import math
import os
import time
from collections import Counter

import numpy as np

from . import _lib
from . import convention_osd as cnv_OSD
from . import globalmap as GL
from ._osd_common import (batch_osd, collect_first_rows, front_batch, full_gf2elim, identify_mrb,  # noqa: F401
                          miracle_view, swapped_info)

CHUNK = 8192


def binomial_coefficient(n, k):
    return math.factorial(n) // (math.factorial(k) * math.factorial(n - k))


def _binom_cdf(b, n, p):
    q = 1.0 - p
    t = q ** n
    acc = t
    for i in range(int(b)):
        t = t * ((n - i) / (i + 1)) * (p / q)
        acc += t
    return acc


def calculate_two_thresholds(pt):
    """(p_t_suc, p_t_pro, N_max) for a mean MRB bit-error probability (pb_testing.py:485-500)."""
    code = GL.get_map('code_parameters')
    order_limit = GL.get_map('order_limit')
    niu = _binom_cdf(order_limit, code.k, float(pt))
    comb_sum = sum(binomial_coefficient(code.k, i) for i in range(order_limit + 1))
    return 0.99 * niu, 0.002 * math.sqrt((1 - niu) / comb_sum), comb_sum


def _cut(fails, threshold):
    """Number of leading frames the reference's loop visits: up to and including the frame on which
    the running failure count reaches ``threshold`` (:174)."""
    c = np.cumsum(fails)
    hit = np.flatnonzero(c >= threshold)
    return int(hit[0]) + 1 if hit.size else len(fails)


def _mrb_errors(ys, labs):
    _, perm, _ = front_batch(ys)
    p = perm.cpu().numpy().astype(np.int64)[:, :64]
    hard = (np.take_along_axis(ys, p, axis=1) <= 0).astype(np.int64)
    return ((hard + np.take_along_axis(labs, p, axis=1)) % 2).sum(axis=1)


def pb_osd(snr, selected_ds):
    start_time = time.process_time()
    order_limit = GL.get_map('order_limit')
    threshold = GL.get_map('termination_num_threshlod', 100)
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
        teps_size = boundaries[-1]
        total = ok + bad
        FER = round(bad / max(total, 1), 4)
        T2 = time.process_time()
        print('\nFor CNV-OSD %.1fdB (order_limit:%d) ' % (snr, order_limit) + ':\n')
        print('----> S:' + str(ok) + ' F:' + str(bad) + '\n')
        print('Distribution of phases:' + str(convention_counter) + '\n')
        print('FER:' + str(FER) + ' Average TEPs size:', teps_size, '\n')
        os.makedirs(logdir, exist_ok=True)
        with open(logdir + 'CNV-OSD-order-' + str(order_limit) + '.txt', 'a+') as f:
            f.write('\nFor CNV-OSD %.1fdB (order_limit:%d) summary:\n' % (snr, order_limit))
            f.write('----> S:' + str(ok) + ' F:' + str(bad) + '\n')
            f.write('Distribution of phases:' + str(convention_counter) + '\n')
            f.write('FER:' + str(FER) + ' Average TEPs size:' + str(teps_size) + '\n')
            f.write(f'Running time:{T2 - start_time} seconds with mean time {(T2 - start_time) / max(total, 1):.4f}!\n')
        summary['convention_osd'] = dict(S=ok, F=bad, FER=FER, teps=teps_size, phases=dict(convention_counter))

    if GL.get_map('pb_osd', False):
        N_max = sum(binomial_coefficient(64, i) for i in range(order_limit + 1))
        fails, nteps, aux = [], [], []
        for s in range(0, len(ys), CHUNK):                 # chunked so the stop rule can end the run early
            r = batch_osd(ys[s:s + CHUNK], labs[s:s + CHUNK], order_limit, _lib.OSD_PB, snr_db=float(snr))
            fails.append(~r['correct']); nteps.append(r['ntep']); aux.append(r['aux'])
            if np.concatenate(fails).sum() >= threshold:
                break
        fails = np.concatenate(fails) if fails else np.zeros(0, bool)
        nteps = np.concatenate(nteps) if nteps else np.zeros(0, np.int32)
        aux = np.concatenate(aux) if aux else np.zeros((0, 4), np.int32)
        n = _cut(fails, threshold)
        fail_sum, correct_sum = int(fails[:n].sum()), int(n - fails[:n].sum())
        stopped = aux[:n, 3] != 0
        counter_teps_sum = int(np.where(stopped, nteps[:n], N_max).sum())      # :152-155
        memory_sum, suc1, suc2 = int(aux[:n, 0].sum()), int(aux[:n, 1].sum()), int(aux[:n, 2].sum())
        actual_size = max(correct_sum + fail_sum, 1)
        FER = round(fail_sum / actual_size, 4)
        average_size = round(counter_teps_sum / actual_size, 5)
        average_num_memory = round(memory_sum / actual_size, 5)
        a1, a2 = round(suc1 / actual_size, 5), round(suc2 / actual_size, 5)
        T2 = time.process_time()
        print('\nFor PB-OSD %.1fdB (order_limit:%d) ' % (snr, order_limit) + ':\n')
        print('----> S:' + str(correct_sum) + ' F:' + str(fail_sum) + '\n')
        print(f'FER:{FER:.4f} Average TEPs:{average_size:.2f} Maintained_list_len:{average_num_memory:.2f} Average_suc: {a1:.2f}/{a2:.2f}')
        print(f'Running time:{T2 - start_time} seconds with mean time {(T2 - start_time) / actual_size:.4f}!')
        os.makedirs(logdir, exist_ok=True)
        with open(logdir + 'PB-OSD-order-' + str(order_limit) + '.txt', 'a+') as f:
            f.write('\nFor PB-OSD %.1fdB (order_limit:%d) summary:\n' % (snr, order_limit))
            f.write(f'--> S/F:{correct_sum}/{fail_sum}\n')
            f.write(f'FER:{FER:.5f} Average TEPs:{average_size:.2f} Maintained_list_len:{average_num_memory:.2f} Average_suc: {a1:.2f}/{a2:2f}\n')
            f.write(f'Running time:{T2 - start_time} seconds with mean time {(T2 - start_time) / actual_size:.4f}!\n')
        summary['pb_osd'] = dict(S=correct_sum, F=fail_sum, FER=FER, average_teps=average_size,
                                 maintained_list=average_num_memory, suc=(a1, a2), frames=n)
    return summary
