#!/usr/bin/env python3
"""Timing of the PB-OSD routes on chosen frame sets (GPU box): where the time of the workgroup kernel goes.

    python scripts/bench_pb_paths.py [--order 3]

Sets: NMS-10 failures at 2.5 dB decoded with the true SNR (typical searches), and the same frames decoded with
snr_db = -5 (no rule fires: every frame scans all N_max TEPs -- the latency of a full scan).  The search alone is
timed (ldpc_osd_search on precomputed front-end results), HIP events, median of 5.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from short_ldpc_decoding_osd_amd import Code, _lib  # noqa: E402
from short_ldpc_decoding_osd_amd.runtime import Decoder  # noqa: E402


def frames(dec, B, seed, snr=2.5):
    g = torch.Generator(device=dec.device).manual_seed(seed)
    G = torch.from_numpy(dec.code.G).to(device=dec.device, dtype=torch.float32)
    sigma = float(np.sqrt(1.0 / (2.0 * 0.5 * 10.0 ** (snr / 10.0))))
    msg = torch.randint(0, 2, (B, 64), device=dec.device, generator=g).to(torch.float32)
    cw = (msg @ G).remainder_(2)
    return ((1 - 2 * cw) * (1 + sigma * torch.randn((B, 128), device=dec.device, generator=g))).contiguous()


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--order", type=int, default=3)
    args = ap.parse_args()
    dec = Decoder(Code(), 0)
    y = frames(dec, 1 << 17, 1)
    res = dec.nms(y, 10, 0.669435)
    index, count = dec.compact(res["fail"])
    nf = int(count.cpu()[0])
    yf = y[index[:nf].long()].contiguous()
    perm, parity, _ = dec.osd_front(yf)
    out = []
    for name, F, snr, path in [] if os.environ.get("LDPC_S2_GRID_MULT") else [("typical staged", nf, 2.5, None), ("typical all-through-workgroup-kernel", nf, 2.5, "block"),
                               ("typical 8192 workgroup-kernel", 8192, 2.5, "block"), ("typical 1024 workgroup-kernel", 1024, 2.5, "block"),
                               ("full scans x256", 256, -5.0, "block"), ("full scans x1024", 1024, -5.0, "block"),
                               ("full scans x4096", 4096, -5.0, "block"), ("full scans x64 list replay", 64, -5.0, "replay")]:
        aux = torch.zeros((F, 4), dtype=torch.int32, device=dec.device)
        p = dec.osd_params(args.order, _lib.OSD_PB, snr_db=snr, aux=aux, pb_path=path)
        o = {}

        def run():
            o.update(dec.osd_search(yf[:F], perm[:F], parity[:F], p, out=o if o else None))
        ms = timed(run)
        nt = o["ntep"].cpu().numpy()
        out.append(dict(set=name, frames=F, ms=ms, us_per_frame=1e3 * ms / F, mean_teps=float(nt.mean()), max_teps=int(nt.max()),
                        stop_hist=np.bincount(aux[:, 3].cpu().numpy(), minlength=3).tolist()))
        print(json.dumps(out[-1]), flush=True)
    for order in (2,):
        o = {}
        p = dec.osd_params(order)
        ms_front = timed(lambda: dec.osd_front(yf, out=(perm, parity, torch.empty(nf, dtype=torch.int32, device=dec.device))))
        ms_s2 = timed(lambda: o.update(dec.osd_search(yf, perm, parity, p, out=o if o else None)))
        print(json.dumps(dict(set="conventional", frames=nf, front_ms=ms_front, search2_ms=ms_s2)), flush=True)


if __name__ == "__main__":
    main()
