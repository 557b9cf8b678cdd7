"""-m gpu: H-form OSD primitives of the DL-OSD stage (ldpc_hosd_front / ldpc_hosd_search and the
``ordered_statistics_decoding.osd`` mirror) against the NumPy oracle (np_oracle.hosd_*) and against the
reference's own gf2elim outputs (tests/golden/gf2elim_ccsds_hform.npz).  Integer results (sort order,
index bookkeeping, M, argmins, codewords) must be identical; the metrics follow the canonical float
order of np_oracle.hosd_cost and must be bit-identical too."""
import os

import numpy as np
import pytest
import torch

from oracle import np_oracle
from tests.gpu_util import pack_np, to_dev, words_np

pytestmark = pytest.mark.gpu

PATH = [[0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0], [0, 1, 0, 0, 0, 0], [0, 0, 1, 0, 0, 0], [1, 1, 0, 0, 0, 0],
        [0, 0, 0, 1, 0, 0], [0, 2, 0, 0, 0, 0], [0, 0, 0, 0, 1, 0], [0, 1, 1, 0, 0, 0], [0, 0, 0, 0, 0, 1],
        [2, 0, 0, 0, 0, 0], [1, 0, 1, 0, 0, 0], [0, 0, 2, 0, 0, 0], [1, 1, 1, 0, 0, 0], [0, 1, 0, 1, 0, 0]]


@pytest.fixture(scope="module")
def dec():
    from short_ldpc_decoding_osd_amd import Code
    from short_ldpc_decoding_osd_amd.runtime import Decoder
    return Decoder(Code())


@pytest.fixture(scope="module")
def blocks():
    _, b = np_oracle.segment_boundaries(64, 6)
    ranges = [range(int(b[i]), int(b[i + 1])) for i in range(6)]
    return [np_oracle.error_pattern_gen(p, ranges, 64) for p in PATH]


def _frames(code, n, seed, refine=True):
    """Channel values y (metric input), a second, different value vector x that orders the positions
    (stand-in for the CNN-refined LLRs: the NMS posterior), labels."""
    rng = np.random.default_rng(seed)
    y, cw = np_oracle.make_frames(code.G, 2.5, n, rng)[:2]
    y = y.astype(np.float32)
    x = np_oracle.nms_sparse(y, code.H, 4, np.float32(0.669435))[-1].astype(np.float32) if refine else y.copy()
    return x, y, cw


def test_front_matches_reference_golden(dec, golden_dir):
    g = np.load(os.path.join(golden_dir, "gf2elim_ccsds_hform.npz"))
    red = np.unpackbits(g["reduced"], axis=2)[:, :, :128]
    lri, uidx, M, ns = dec.hosd_front(to_dev(g["y"], dec))
    lri, uidx, ns = lri.cpu().numpy(), uidx.cpu().numpy(), ns.cpu().numpy()
    Mb = np.unpackbits(M.cpu().numpy().view(np.uint8).reshape(-1, 64, 8), axis=2, bitorder="little")
    assert np.array_equal(lri, g["perm"])                       # ascending stable sort
    assert np.array_equal(ns, g["nswaps"])
    for i in range(len(ns)):
        idx = np.arange(128)
        for a, b in g["swaps"][i][: g["nswaps"][i]]:
            idx[a], idx[b] = idx[b], idx[a]
        sw = np.argsort(idx[64:], kind="stable")
        assert np.array_equal(uidx[i], np.concatenate([idx[:64], idx[64:][sw]])), i
        assert np.array_equal(Mb[i], red[i][:, 64:][:, sw]), i   # the reference's reduced matrix, MRB columns sorted


def test_front_and_search_match_oracle(dec, np_code, blocks):
    x, y, cw = _frames(np_code, 160, 11)
    x[3, 10] = x[3, 77] = 0.0                                   # exact ties incl. zeros
    x[4, :] = np.float32(0.5) * np.sign(x[4, :])                # every key equal
    y[5, 20] = 0.0                                              # zero weight in the metric
    x[6] *= np.float32(40.0)                                    # other scalings of the ordering values
    x[7] *= np.float32(1e-3)
    x[8, 77] = np.float32(1e30)                                 # an outlier: every other value in one sort bucket
    x[9, :] = 0.0
    teps = np.concatenate([_tab(E) for E in blocks])
    off = np.insert(np.cumsum([len(E) for E in blocks]), 0, 0).astype(np.int32)
    front = dec.hosd_front(to_dev(x, dec))
    out = dec.hosd_search(to_dev(x, dec), to_dev(y, dec), front, to_dev(teps, dec), to_dev(off, dec),
                          label_bits=to_dev(pack_np(cw).view(np.int64), dec))
    torch.cuda.synchronize()
    lri, uidx, M, ns = (t.cpu().numpy() for t in front)
    Mb = np.unpackbits(M.view(np.uint8).reshape(-1, 64, 8), axis=2, bitorder="little")
    bmin, barg = out["block_min"].cpu().numpy(), out["block_arg"].cpu().numpy()
    cwg = np.unpackbits(words_np(out["cw"]).view(np.uint8).reshape(-1, 16), axis=1, bitorder="little")
    for f in range(x.shape[0]):
        r = np_oracle.hosd_frame(x[f], y[f], cw[f], np_code.H, blocks)
        assert np.array_equal(lri[f], r["lri"]) and np.array_equal(uidx[f], r["uidx"]), f
        assert np.array_equal(Mb[f], r["M"]) and ns[f] == len(r["swaps"]), f
        assert np.array_equal(bmin[f].view(np.uint32), r["block_min"].view(np.uint32)), f
        assert np.array_equal(barg[f], r["block_arg"]), f
        assert out["truth"][f].item() == r["truth"] and out["metric"][f].item() == r["metric"], f
        assert out["best"][f].item() == r["best_index"] and np.array_equal(cwg[f], r["codeword"]), f
        assert not (np_code.H.dot(cwg[f]) % 2).any()


def _tab(E):
    from short_ldpc_decoding_osd_amd.ordered_statistics_decoding import _teps_from_matrix
    return _teps_from_matrix(E)


def test_edge_cases(dec, np_code, blocks):
    e = dec.empty((0, 128), torch.float32)
    front = dec.hosd_front(e)
    assert front[0].shape == (0, 128)
    teps = to_dev(_tab(blocks[1]), dec)
    off = to_dev(np.array([0, 0, 1, 1], dtype=np.int32), dec)   # empty blocks around a one-TEP block
    x, y, cw = _frames(np_code, 3, 2, refine=False)
    fr = dec.hosd_front(to_dev(x, dec))
    out = dec.hosd_search(to_dev(x, dec), to_dev(y, dec), fr, teps, off)
    bm, ba = out["block_min"].cpu().numpy(), out["block_arg"].cpu().numpy()
    assert np.all(np.isinf(bm[:, 0])) and np.all(np.isinf(bm[:, 2])) and np.all(ba[:, 0] == -1) and np.all(ba[:, 1] == 0)
    assert np.array_equal(out["best"].cpu().numpy(), [0, 0, 0])
    from short_ldpc_decoding_osd_amd import _lib
    with pytest.raises(_lib.LdpcError):                         # truth without labels is an argument error
        _lib.check(dec.L.ldpc_hosd_search(dec._ctx, x.ctypes.data, y.ctypes.data, 1, 1, 1, 1, 1, 1, 1, None, 1, None, 1,
                                          None, None, None, None))


def test_full_scale_properties(dec, np_code, blocks):
    """65 536 frames: every winner is a codeword, never worse than the order-0 candidate; with the label's
    error pattern inside the scanned blocks the winner's metric is <= the label's; permutation outputs are
    permutations."""
    rng = np.random.default_rng(7)
    B = 65536
    y, cw = np_oracle.make_frames(np_code.G, 2.5, B, rng)[:2]
    y = y.astype(np.float32)
    yd = to_dev(y, dec)
    teps = to_dev(np.concatenate([_tab(E) for E in blocks]), dec)
    off = to_dev(np.insert(np.cumsum([len(E) for E in blocks]), 0, 0).astype(np.int32), dec)
    front = dec.hosd_front(yd)
    out = dec.hosd_search(yd, yd, front, teps, off, label_bits=to_dev(pack_np(cw).view(np.int64), dec))
    torch.cuda.synchronize()
    assert (front[3] >= 0).all()
    comp = torch.gather(front[0].long(), 1, front[1].long())
    assert (comp.sort(dim=1).values == torch.arange(128, device=dec.device)[None]).all()
    bits = dec.unpack_bits(out["cw"], dtype=torch.uint8).float()
    Ht = torch.from_numpy(np_code.H.T.astype(np.float32)).to(dec.device)
    assert ((bits @ Ht) % 2 == 0).all()
    bm = out["block_min"]
    assert (out["metric"] == bm.min(dim=1).values).all() and (out["metric"] <= bm[:, 0]).all()
    hit = out["metric"] == out["truth"]
    lab = torch.from_numpy(cw.astype(np.float32)).to(dec.device)
    assert ((bits == lab).all(dim=1) == hit).all()              # equal metric <=> the label itself (ties have measure 0)
    assert (out["metric"][~hit] < out["truth"][~hit]).sum() + hit.sum() > 0.9 * B   # most misses are ML-better or hits
    assert hit.float().mean() > 0.9


def test_mirror_sliding_osd(dec, np_code, blocks):
    from short_ldpc_decoding_osd_amd import Code, ordered_statistics_decoding as osd_mod
    from short_ldpc_decoding_osd_amd import globalmap as GL
    T = 4
    GL.set_map('code_parameters', Code())
    GL.set_map('num_iterations', T)
    for k, v in dict(segment_num=6, threshold_sum=3, decoding_length=len(PATH), sliding_win_width=5, soft_margin=0.9).items():
        GL.set_map(k, v)
    x, y, cw = _frames(np_code, 40, 21)
    traj = np_oracle.nms_sparse(y, np_code.H, T, np.float32(0.669435))
    input_list = np.stack(traj, axis=1).reshape(-1, 128).astype(np.float32)    # (T+1) rows per frame, row 0 = channel
    inst = osd_mod.osd(Code())
    tep_info = osd_mod.generate_teps(inst, PATH)
    assert all(np.array_equal(a, b) for a, b in zip(tep_info[0], blocks))

    def fcn(v):                                                 # stand-in for Predict_outlier_light: stop when the
        v = np.asarray(v).reshape(-1)                           # window's best is far below its median
        p = 1.0 / (1.0 + np.exp(-(v[2] - v[0] - 3.0)))
        return np.array([1 - p, p])

    s, f_, wins, cplx = inst.sliding_osd(fcn, input_list, x, cw, tep_info)
    want = [np_oracle.sliding_window_decide(np_oracle.hosd_frame(x[i], y[i], cw[i], np_code.H, blocks)["block_min"],
                                            np_oracle.hosd_frame(x[i], y[i], cw[i], np_code.H, blocks)["truth"], fcn, 5, 0.9,
                                            tep_info[1]) for i in range(len(x))]
    assert s == sum(w[0] for w in want) and f_ == len(x) - s
    assert wins == sum(w[1] for w in want) and cplx == sum(w[2] for w in want)
    assert np.array_equal(inst.last["success"], [w[0] for w in want])
    assert 0 < wins < len(x) * (len(PATH) - 5 + 1)              # the stop rule fired for some frames only
    # per-frame functions on frame 0, as the reference's loop would call them
    oH, oin, oorig, olab = inst.check_matrix_reorder(input_list, x, cw)
    idx_l, M_l, len_l, pos_l = inst.identify_mrb(oH[:2])
    r0 = np_oracle.hosd_frame(x[0], y[0], cw[0], np_code.H, blocks)
    assert np.array_equal(idx_l[0], r0["uidx"]) and np.array_equal(M_l[0], r0["M"])
    assert len_l[0] == int((r0["uidx"][64:] < 64).sum()) and pos_l[0].shape == (64,)
    o_orig = oorig[0][0][idx_l[0]]
    o_in = oin[0][idx_l[0]]
    got = inst.acquire_min(blocks[4], np.where(o_in > 0, 0, 1)[64:], M_l[0], np.where(o_orig > 0, 0, 1), np.abs(o_orig))
    assert got == r0["block_min"][4]
    assert np.array_equal(inst.unpack_codewords(inst.last["cw"])[0], r0["codeword"])
