#!/usr/bin/env python3
"""Golden-vector generator -- TEST INFRASTRUCTURE, runs only in the build container.

Imports the hot-path modules of the reference that are importable without TensorFlow --
``LDPC_128/Ldpc_128_testing/fill_matrix_info.py`` and ``LDPC_128/Testing_data_gen_128/data_generating.py``
(with that directory's own ``globalmap.py``), both NumPy only -- from /root/reference and records their
outputs as small fixtures under tests/golden/:

  code_<name>.npz       H, G, k for each code            (Code.load_code :70-129,
                                                           Code.generator_matrix :44-69)
  gf2elim_ccsds.npz     reliability-permuted copies of G  -> Code.gf2elim output + recorded
                        column exchanges                  (Code.gf2elim :7-42, the routine
                        that PB_OSD/pb_testing.py:231-266 ``full_gf2elim`` repeats verbatim)

  gf2elim_ccsds_hform.npz  ascending-|y| column orders of H -> Code.gf2elim output + exchanges (the
                        same routine as DL_OSD_Testing_serial/ordered_statistics_decoding.py:222-257,
                        as ``osd.identify_mrb`` :43-80 applies it to the permuted parity-check matrix)

  testgen_ccsds.npz     testing_data_generating(code, SNR, frames) under np.random.seed(s) for three (s, SNR):
                        the reference's frames and labels   (Testing_data_gen_128/data_generating.py:13-51)

Only DATA is written (inputs and the reference's outputs); no reference source travels.
The reference tree is imported with bytecode writing disabled so nothing is written there.

    python oracle/gen_golden.py            # rewrites tests/golden/*.npz
    python oracle/gen_golden.py testgen    # only tests/golden/testgen_ccsds.npz
"""
import contextlib
import io
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference/LDPC_128/Ldpc_128_testing"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")

CODES = {
    # fixture name -> alist (data files of the reference)
    "ccsds_128_64": os.path.join(REF, "CCSDS_ldpc_n128_k64.alist"),
    "array_121_60": os.path.join(REF, "ArrayCode_N121_K60_r0.50.alist"),
    "ldpc_96_48": "/root/reference/LDPC_128/Ldpc_128_training/LDPC_N96_K48_P8_set0_dmin10.alist",
}


TESTGEN_DIR = "/root/reference/LDPC_128/Testing_data_gen_128"
TESTGEN_CASES = ((20241020, 2.5, 96), (7, 1.0, 64), (123456, 3.5, 64))   # (np.random.seed, SNR dB, frames)


def gen_testgen(code):
    """The reference's own test-frame generator (a10) on its global, here seeded, NumPy RNG."""
    sys.path.insert(0, TESTGEN_DIR)
    for name in ("globalmap", "data_generating"):
        sys.modules.pop(name, None)
    import data_generating as refgen   # the reference's module; imports the globalmap.py next to it
    import globalmap as refgl
    refgl.set_map("Rayleigh_fading", False)
    refgl.set_map("ALL_ZEROS_CODEWORD_TESTING", False)
    out = {}
    for i, (seed, snr, frames) in enumerate(TESTGEN_CASES):
        np.random.seed(seed)
        data, labels = refgen.testing_data_generating(code, snr, frames)
        assert data.dtype == np.float64 and data.shape == (frames, 128)
        out[f"data{i}"], out[f"labels{i}"] = data, labels.astype(np.uint8)
    out["cases"] = np.array(TESTGEN_CASES, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "testgen_ccsds.npz"), **out)
    sys.path.remove(TESTGEN_DIR)
    for name in ("globalmap", "data_generating"):
        sys.modules.pop(name, None)
    print("testgen cases", TESTGEN_CASES)


def main():
    sys.path.insert(0, REF)
    import fill_matrix_info as ref  # the reference's own module
    if sys.argv[1:] == ["testgen"]:
        with contextlib.redirect_stdout(io.StringIO()):
            code = ref.Code(CODES["ccsds_128_64"])
        os.makedirs(OUT, exist_ok=True)
        gen_testgen(code)
        return

    os.makedirs(OUT, exist_ok=True)
    codes = {}
    for name, path in CODES.items():
        with contextlib.redirect_stdout(io.StringIO()):
            code = ref.Code(path)
        codes[name] = code
        np.savez_compressed(
            os.path.join(OUT, f"code_{name}.npz"),
            H=code.H.astype(np.uint8), G=code.G.astype(np.uint8), k=np.int64(code.k),
            max_chk_degree=np.int64(code.max_chk_degree))
        print(name, "H", code.H.shape, "G", code.G.shape, "k", code.k)

    # ---- GE cases on the CCSDS code: G with columns in descending-|y| order ------------
    code = codes["ccsds_128_64"]
    G = code.G
    k, n = G.shape
    rng = np.random.default_rng(20241020)
    cases = 384
    ys = np.empty((cases, n), dtype=np.float32)
    for i in range(cases):
        snr = (1.0, 2.5, 3.5)[i % 3]
        sigma = np.sqrt(1.0 / (2.0 * (k / n) * 10 ** (snr / 10)))
        cw = rng.integers(0, 2, size=k).dot(G) % 2
        ys[i] = np.where(cw == 0, 1, -1) * rng.normal(1.0, sigma, size=n)
    # a few adversarial orders: long column-exchange chains
    ys[-1] = np.linspace(2.0, 0.1, n)                       # identity order: parity part first
    ys[-2] = np.linspace(0.1, 2.0, n)                       # reversed order
    ys[-3] = np.concatenate([np.linspace(0.1, 1.0, 64), np.linspace(3.0, 2.0, 64)])
    perms = np.argsort(-np.abs(ys), axis=1, kind="stable")
    red = np.empty((cases, k, n), dtype=np.uint8)
    swaps = np.full((cases, 64, 2), -1, dtype=np.int16)
    nswaps = np.zeros(cases, dtype=np.int16)
    for i in range(cases):
        M, rec = code.gf2elim(np.copy(G[:, perms[i]]))
        assert M.shape == (k, n)
        red[i] = M
        nswaps[i] = len(rec)
        for t, (a, b) in enumerate(rec):
            swaps[i, t] = (a, b)
    np.savez_compressed(
        os.path.join(OUT, "gf2elim_ccsds.npz"),
        y=ys, perm=perms.astype(np.int16), reduced=np.packbits(red, axis=2),
        swaps=swaps, nswaps=nswaps)
    print("gf2elim cases", cases, "swaps mean %.2f max %d" % (nswaps.mean(), nswaps.max()))

    # ---- H-form cases: H with columns in ascending-|y| order (DL-OSD stage) ---------------
    H = code.H
    m = H.shape[0]
    hcases = 192
    hy = np.concatenate([ys[:hcases - 3], ys[-3:]])
    hperms = np.argsort(np.abs(hy), axis=1, kind="stable")
    hred = np.empty((hcases, m, n), dtype=np.uint8)
    hswaps = np.full((hcases, 64, 2), -1, dtype=np.int16)
    hns = np.zeros(hcases, dtype=np.int16)
    for i in range(hcases):
        M, rec = code.gf2elim(np.copy(H[:, hperms[i]]))
        assert M.shape == (m, n)
        hred[i] = M
        hns[i] = len(rec)
        for t, (a, b) in enumerate(rec):
            hswaps[i, t] = (a, b)
    np.savez_compressed(
        os.path.join(OUT, "gf2elim_ccsds_hform.npz"),
        y=hy, perm=hperms.astype(np.int16), reduced=np.packbits(hred, axis=2), swaps=hswaps, nswaps=hns)
    print("H-form gf2elim cases", hcases, "swaps mean %.2f max %d" % (hns.mean(), hns.max()))
    gen_testgen(code)


if __name__ == "__main__":
    main()
