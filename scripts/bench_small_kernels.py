#!/usr/bin/env python3
"""Stand-alone timing of the small streaming kernels (GPU box): compaction alone, back to back, versus inside the step."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from short_ldpc_decoding_osd_amd import Code
from short_ldpc_decoding_osd_amd.runtime import Decoder
dec = Decoder(Code(), 0)
B = 131072
flag = (torch.rand(B, device=dec.device) < 0.25).to(torch.uint8)
index, count = dec.compact(flag)
def timed(fn, n=200):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
print(json.dumps(dict(compact_back_to_back_us=timed(lambda: dec.compact(flag, index, count)))))
big = torch.empty(64 << 20, dtype=torch.uint8, device=dec.device)
def dirty_then_compact():
    big.fill_(1)          # 64 MiB of dirty lines right before
    dec.compact(flag, index, count)
t_both = timed(dirty_then_compact, 50); t_fill = timed(lambda: big.fill_(1), 50)
print(json.dumps(dict(fill64MiB_us=t_fill, fill_then_compact_us=t_both, compact_after_dirty_us=t_both - t_fill)))
