"""python scripts/pb_order_experiment.py [snr]: does the ORDER of the frames in a PB-OSD launch matter?  One 131 072-frame batch
of NMS failures, PB-OSD order 3 through ldpc_osd_search, HIP-event timed: the batch as it comes, sorted by the search length
(longest first / shortest first; an oracle ordering no product path has), and by predictors the singles kernel could compute."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import torch
from short_ldpc_decoding_osd_amd import Code, _lib
from short_ldpc_decoding_osd_amd.runtime import Decoder
import bench
snr = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
only = sys.argv[2] if len(sys.argv) > 2 else None        # one ordering only (under rocprofv3: kernel stats of that ordering)
dec = Decoder(Code(), 0)
y, _ = bench.make_frames(dec, 1 << 17, 1, snr_db=snr)
res = dec.nms(y, 10, 0.669435)
index, count = dec.compact(res["fail"]); nf = int(count.cpu()[0])
yf = y[index[:nf].long()].contiguous()
perm, parity, _ = dec.osd_front(yf)
p = dec.osd_params(3, _lib.OSD_PB, snr_db=snr)
ref = dec.osd_search(yf, perm, parity, p)
torch.cuda.synchronize()
ntep = ref["ntep"].long()
print(f"snr {snr}: {nf} frames, mean TEPs {ntep.float().mean().item():.1f}, max {ntep.max().item()}, "
      f"> 4096: {(ntep > 4096).sum().item()}, > 512: {(ntep > 512).sum().item()}", flush=True)

def timed(order, name, reps=20):
    if only is not None and not name.startswith(only): return
    yo, po, qo = yf[order].contiguous(), perm[order].contiguous(), parity[order].contiguous()
    out = dec.osd_search(yo, po, qo, p); torch.cuda.synchronize()
    assert torch.equal(out["ntep"].long(), ntep[order]), name
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): dec.osd_search(yo, po, qo, p, out=out)
    e1.record(); torch.cuda.synchronize()
    print(f"  {name:34s} {e0.elapsed_time(e1) / reps * 1e3:8.1f} us", flush=True)

ident = torch.arange(nf, device=yf.device)
# predictor: reliability of the least reliable basis positions (what the singles kernel sees first)
w = yf.abs().gather(1, perm.long())            # |y| in the elimination order: basis = first 64
soft = w[:, :64].sort(dim=1).values[:, :8].sum(dim=1)
for rep in range(2):
    timed(ident, "as it comes")
    timed(torch.argsort(ntep, descending=True, stable=True), "longest search first (oracle)")
    timed(torch.argsort(ntep, descending=False, stable=True), "shortest search first (oracle)")
    timed(torch.randperm(nf, device=yf.device), "random")
    timed(torch.argsort(soft, descending=False, stable=True), "weakest basis first (predictor)")
