"""Conventional order-2 OSD of one batch's NMS failures, two routes, HIP-event timed (GPU box):
   two kernels through caller buffers (ldpc_osd_front + ldpc_osd_search) / ldpc_osd_decode (one fused kernel)."""
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
p = dec.osd_params(2)
front = dec.osd_front(y, index=idx)
o1 = dec.osd_search(y, front[0], front[1], p, index=idx)
o2 = dec.osd_decode(y, 2, index=idx)
torch.cuda.synchronize()
for k in ("cw", "metric", "best", "ntep"):
    assert torch.equal(o1[k], o2[k]), k
def timed(fn, reps=40):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for rep in range(3):
    t2 = timed(lambda: (dec.osd_front(y, index=idx, out=front), dec.osd_search(y, front[0], front[1], p, index=idx, out=o1)))
    t1 = timed(lambda: dec.osd_decode(y, 2, index=idx, out=o2))
    print(f"{nf} frames: front + scan {t2:.1f} us, fused {t1:.1f} us", flush=True)
