"""OSD front-end functions shared by the ``pb_testing`` and ``fs_testing`` mirrors (the reference
carries identical copies: PB_OSD/pb_testing.py:231-320 == FS_OSD/fs_testing.py:233-322)."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from . import globalmap as GL
from .runtime import default_decoder


def _dec():
    return default_decoder(GL.get_map('code_parameters'))


def _pack_rows(M):
    """[..., 64, 128] 0/1 -> [..., 64, 2] int64 words (bit c of a row at word c // 64)."""
    M = np.ascontiguousarray(M, dtype=np.uint8)
    return np.packbits(M, axis=-1, bitorder="little").view(np.int64).reshape(M.shape[:-1] + (2,))


def _unpack_rows(words, ncols=128):
    b = np.unpackbits(np.ascontiguousarray(words).view(np.uint8), axis=-1, bitorder="little")
    return b.reshape(words.shape[:-1] + (-1,))[..., :ncols]


def full_gf2elim(M):
    """``full_gf2elim`` (pb_testing.py:231-266): (reduced M, [(j, col), ...]).  64 x 128 matrices (the
    per-frame case) run on the device (``ldpc_osd_ge``); other shapes use the library's host
    elimination (the one-time code-construction path)."""
    M = np.asarray(M)
    code = GL.get_map('code_parameters')
    if M.shape == (64, 128) and code.k == 64 and code.check_matrix_column == 128:
        dec = _dec()
        red, swaps, ns = dec.osd_ge(torch.from_numpy(_pack_rows(M)[None]).to(dec.device))
        n = int(ns.cpu()[0])
        R = _unpack_rows(red.cpu().numpy()[0]).astype(M.dtype)
        if M.flags.writeable:
            M[...] = R          # the reference works in place on its argument
        return R, [(int(a), int(b)) for a, b in swaps.cpu().numpy()[0, :n]]
    return code.gf2elim(M)


def identify_mrb(order_inputs, order_G):
    """``identify_mrb`` (pb_testing.py:268-304) for an already reliability-ordered G.
    The elimination runs on the device; the index bookkeeping of :276-302 (a 128-entry permutation)
    is replayed here.  ``swapped_info`` -- what the drivers call -- does all of it in one kernel."""
    code = GL.get_map('code_parameters')
    k, n = code.k, code.check_matrix_column
    R, swaps = full_gf2elim(np.array(order_G, dtype=np.int64))
    idx = np.arange(n)
    for a, b in swaps:
        idx[a], idx[b] = idx[b], idx[a]
    sm, sl = np.argsort(idx[:k], kind="stable"), np.argsort(idx[k:], kind="stable")
    updated_G = np.concatenate([np.identity(k, dtype=np.int32), R[:, k:][:, sl][sm, :].astype(np.int32)], axis=1)
    return updated_G, np.concatenate([idx[:k][sm], idx[k:][sl]])


def front_batch(inputs):
    """[F,128] channel values -> (perm [F,128] int64, P' rows [F,64] packed int64) on the device."""
    dec = _dec()
    y = torch.from_numpy(np.ascontiguousarray(inputs, dtype=np.float32)).to(dec.device)
    perm, parity, _ = dec.osd_front(y)
    return y, perm, parity


def swapped_info(inputs, labels):
    """``swapped_info`` (pb_testing.py:306-320): -> (updated_inputs, updated_labels, reduced_G) with
    reduced_G = [I | P'] int32 [64,128].  One device call (sort + elimination + bookkeeping)."""
    inputs = np.asarray(inputs, dtype=np.float32)
    _, perm, parity = front_batch(inputs[None])
    p = perm.cpu().numpy()[0].astype(np.int64)
    P = _unpack_rows(parity.cpu().numpy()[0][:, None], 64).reshape(64, 64)
    reduced_G = np.concatenate([np.identity(64, dtype=np.int32), P.astype(np.int32)], axis=1)
    return inputs[p], np.asarray(labels)[p], reduced_G


def miracle_view(updated_inputs, updated_labels, reduced_G, counter_stat):
    """Genie statistic: hard-decision errors inside the MRB (pb_testing.py:502-509)."""
    code = GL.get_map('code_parameters')
    hard = np.where(np.asarray(updated_inputs) > 0, 0, 1)
    mrb_error_num = int(((hard[:code.k] + np.asarray(updated_labels)[:code.k]) % 2).sum())
    counter_stat.update([mrb_error_num])
    return counter_stat, mrb_error_num


def primed_search(updated_inputs, reduced_G, order, algo, **kw):
    """Run a search kernel on primed-domain inputs (perm = identity), as the reference's per-frame
    functions receive them.  Returns the result dict of ``Decoder.osd_search`` moved to NumPy."""
    dec = _dec()
    y = torch.from_numpy(np.ascontiguousarray(np.asarray(updated_inputs, dtype=np.float32)[None])).to(dec.device)
    perm = torch.arange(128, dtype=torch.uint8, device=dec.device)[None].contiguous()
    P = np.packbits(np.asarray(reduced_G)[:, 64:].astype(np.uint8), axis=1, bitorder="little").view(np.int64)
    parity = torch.from_numpy(np.ascontiguousarray(P.reshape(1, 64))).to(dec.device)
    aux = torch.zeros((1, 4), dtype=torch.int32, device=dec.device)
    out = dec.osd_search(y, perm, parity, dec.osd_params(order, algo, aux=aux, **kw))
    res = {k: v.cpu().numpy() for k, v in out.items()}
    res["aux"] = aux.cpu().numpy()
    res["codeword"] = _unpack_rows(res["cw"][0][None]).reshape(-1)[:128].astype(np.int32)   # identity perm: primed order
    return res


def collect_first_rows(selected_ds):
    """Row 0 of every (T+1)-row trajectory batch = the channel values (pb_testing.py:71-72)."""
    ys, labs = [], []
    for batch in selected_ds.as_numpy_iterator():
        ys.append(np.asarray(batch[0][0], dtype=np.float32))
        labs.append(np.asarray(batch[1][0], dtype=np.int64))
    if not ys:
        return np.zeros((0, 128), np.float32), np.zeros((0, 128), np.int64)
    return np.stack(ys), np.stack(labs)


def batch_osd(ys, labs, order, algo, **kw):
    """Whole-dataset OSD on the device: returns dict(correct [F] bool, ntep [F], best [F], aux [F,4],
    cw [F,2], metric [F]) in dataset order."""
    dec = _dec()
    F = ys.shape[0]
    y = torch.from_numpy(np.ascontiguousarray(ys)).to(dec.device)
    aux = torch.zeros((max(F, 1), 4), dtype=torch.int32, device=dec.device)
    out = dec.osd_decode(y, order, params=dec.osd_params(order, algo, aux=aux, **kw))
    label_bits = dec.pack_bits(torch.from_numpy(np.ascontiguousarray(labs)).to(dec.device)) if F else None
    correct = (out["cw"] == label_bits).all(dim=1).cpu().numpy() if F else np.zeros(0, bool)
    return dict(correct=correct, ntep=out["ntep"].cpu().numpy(), best=out["best"].cpu().numpy(),
                aux=aux.cpu().numpy()[:F], cw=out["cw"].cpu().numpy(), metric=out["metric"].cpu().numpy())
