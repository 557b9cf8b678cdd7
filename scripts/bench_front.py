"""OSD front-end kernel alone: python scripts/bench_front.py  (33 k NMS failures of one 131 072-frame batch at 2.5 dB; HIP events)"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import torch
from short_ldpc_decoding_osd_amd import Code
from short_ldpc_decoding_osd_amd.runtime import Decoder
import bench
dec = Decoder(Code(), 0)
y, _ = bench.make_frames(dec, 1 << 17, 1)
res = dec.nms(y, 10, 0.669435)
index, count = dec.compact(res["fail"]); nf = int(count.cpu()[0])
idx = index[:nf].contiguous()
out = dec.osd_front(y, index=idx)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    e0.record()
    for _ in range(50):
        dec.osd_front(y, index=idx, out=out)
    e1.record(); torch.cuda.synchronize()
    print(f"osd_front: {nf} frames, {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call", flush=True)
