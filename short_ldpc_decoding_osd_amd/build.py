"""Builds libldpcosd.so in-tree with hipcc for gfx950 (MI355X).  No GPU is needed to build.

    python -m short_ldpc_decoding_osd_amd.build [--force]
"""
from __future__ import annotations

import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libldpcosd.so")
SOURCES = ["ldpc_host.cpp", "ldpc_api.hip", "ldpc_nms.hip", "ldpc_util.hip", "ldpc_osd.hip",
           "ldpc_osd_pb.hip", "ldpc_hosd.hip"]
# -ffp-contract=off: the float order of the NMS / OSD metric is part of the contract (no FMA fusion)
# -fno-slp-vectorize: packed v_pk_add_f32 is no faster than two v_add_f32 on gfx950 and blocks the
#                      fusion of the DPP rotations into their consumers
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, h) for h in ("ldpc_internal.h", "ldpc_wave.h", "ldpc_search.h", "ldpc_osd_state.h")]
    headers.append(os.path.join(HERE, "..", "include", "ldpc_osd.h"))
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc, "-x", "hip", "-c", s, "-o", o] + FLAGS)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + p.stdout + p.stderr)
        return p.stderr

    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as ex:
        for warn in ex.map(run, jobs):
            if warn and verbose:
                print(warn)
    objs = [os.path.join(OBJ, os.path.splitext(s)[0] + ".o") for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
