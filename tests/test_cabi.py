"""CPU: the C-ABI library loads, exports every symbol the header declares, and its host-only
entry points (code construction, TEP tables) reproduce the reference's golden outputs."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_and_binding_agree():
    from short_ldpc_decoding_osd_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "ldpc_osd.h")).read()
    declared = set(re.findall(r"\b(ldpc_[A-Za-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS)
    L = _lib.load()
    for name in declared:
        assert getattr(L, name) is not None
    assert L.ldpc_abi_version() == int(re.search(r"#define LDPC_OSD_ABI_VERSION (\d+)", hdr).group(1)) == 4
    # struct layouts the binding mirrors (a silent drift here corrupts every OSD call)
    import ctypes as C
    assert C.sizeof(_lib.OsdParams) == 48 and _lib.OsdParams.d_aux.offset == 32 and _lib.OsdParams.y_frames.offset == 40
    # ldpc_pb_tuning: thirteen int32 in the header's order (ABI 4)
    m = re.search(r"typedef struct ldpc_pb_tuning \{(.*?)\} ldpc_pb_tuning;", hdr, re.S)
    names = [n.strip() for decl in re.findall(r"int32_t ([^;]+);", m.group(1)) for n in decl.split(",")]
    assert names == [n for n, _ in _lib.PbTuning._fields_] and C.sizeof(_lib.PbTuning) == 4 * len(names) == 52
    assert _lib.PbTuning.t1.offset == 20 and _lib.PbTuning.handoff_maxlen.offset == 48
    assert "getenv" not in re.sub(r'getenv\("LDPC_PB_PROFILE"\)', "", open(os.path.join(ROOT, "short_ldpc_decoding_osd_amd", "csrc", "ldpc_osd_pb.hip")).read())


@pytest.mark.parametrize("name,alist", [
    ("ccsds_128_64", "short_ldpc_decoding_osd_amd/data/CCSDS_ldpc_n128_k64.alist"),
    ("array_121_60", "tests/golden/ArrayCode_N121_K60_r0.50.alist"),
    ("ldpc_96_48", "tests/golden/LDPC_N96_K48_P8_set0_dmin10.alist")])
def test_code_class_matches_reference(name, alist, golden_dir):
    from short_ldpc_decoding_osd_amd import Code
    g = np.load(os.path.join(golden_dir, f"code_{name}.npz"))
    code = Code(os.path.join(ROOT, alist))
    assert np.array_equal(code.H, g["H"]) and np.array_equal(code.G, g["G"])
    assert (code.k, code.max_chk_degree) == (int(g["k"]), int(g["max_chk_degree"]))
    assert (code.check_matrix_row, code.check_matrix_column) == g["H"].shape
    assert np.array_equal(Code(H=g["H"]).G, g["G"])
    assert np.array_equal(code.generator_matrix(g["H"]), g["G"])


def test_host_gf2elim_matches_reference(golden_dir):
    from short_ldpc_decoding_osd_amd import Code
    code = Code()
    g = np.load(os.path.join(golden_dir, "gf2elim_ccsds.npz"))
    red = np.unpackbits(g["reduced"], axis=2)
    for i in range(g["y"].shape[0]):
        M = code.G[:, g["perm"][i].astype(np.int64)].copy()
        R, sw = code.gf2elim(M)
        assert np.array_equal(R, red[i])
        assert sw == [tuple(int(x) for x in r) for r in g["swaps"][i][: g["nswaps"][i]]]


def test_errors_are_reported_not_raised_across_the_abi(tmp_path):
    from short_ldpc_decoding_osd_amd import Code, _lib
    with pytest.raises(_lib.LdpcError, match="cannot open"):
        Code(str(tmp_path / "nope.alist"))
    bad = tmp_path / "bad.alist"
    bad.write_text("4 2\n1 2\n1 1 1 1\n2 2\n1\n9\n1\n2\n")
    with pytest.raises(_lib.LdpcError, match="out of range"):
        Code(str(bad))
    with pytest.raises(_lib.LdpcError):
        Code(H=np.eye(4, dtype=np.int64))          # m == n: no code
    L = _lib.load()
    assert L.ldpc_code_dims(None, None, None, None, None) == -1


def test_tep_table_host():
    from oracle import np_oracle
    from short_ldpc_decoding_osd_amd import _lib
    L = _lib.load()
    for order in range(4):
        bounds = (C.c_int64 * 4)()
        nt = L.ldpc_tep_table(64, order, None, bounds)
        assert list(bounds)[: order + 1] == np_oracle.tep_boundaries(64, order)
        t = np.empty((nt, 3), dtype=np.uint8)
        L.ldpc_tep_table(64, order, t.ctypes.data_as(C.POINTER(C.c_uint8)), None)
        if order <= 2:
            assert [tuple(int(x) for x in r if x != 255) for r in t] == np_oracle.tep_table(64, order)
        else:
            from oracle import c_oracle
            assert np.array_equal(t, c_oracle.tep_table(64, 3))
    assert L.ldpc_tep_table(64, 4, None, None) < 0


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from short_ldpc_decoding_osd_amd import Code, _lib
    from short_ldpc_decoding_osd_amd.runtime import Decoder
    with pytest.raises(_lib.LdpcError, match="no CPU path|no GPU"):
        Decoder(Code())


def test_fs_tep_order_host():
    """ldpc_tep_table_fs == generate_sequential_teps (fs_testing.py:32-49)."""
    from oracle import np_oracle
    from short_ldpc_decoding_osd_amd import _lib
    L = _lib.load()
    lists = np_oracle.fs_tep_lists(64, 2)
    for w in (1, 2):
        n = L.ldpc_tep_table_fs(64, w, None)
        t = np.empty((n, 3), dtype=np.uint8)
        L.ldpc_tep_table_fs(64, w, t.ctypes.data_as(C.POINTER(C.c_uint8)))
        assert [tuple(int(x) for x in r if x != 255) for r in t] == lists[w - 1]
    assert lists[0][:3] == [(63,), (62,), (61,)] and lists[1][:3] == [(62, 63), (61, 63), (60, 63)]
    assert L.ldpc_tep_table_fs(64, 3, None) == 41664 and L.ldpc_tep_table_fs(64, 4, None) < 0
