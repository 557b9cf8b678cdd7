"""LDPC_PB_PROFILE=1 python scripts/pb_profile_run.py [snr [seed]]: in-kernel phase stamps of the PB-OSD kernels (stderr) on one
131 072-frame batch of NMS failures (typical searches) -- diagnostic instantiations, not the product kernels."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch
from short_ldpc_decoding_osd_amd import Code, _lib
from short_ldpc_decoding_osd_amd.runtime import Decoder
sys.path.insert(0, ROOT)
import bench
snr = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
dec = Decoder(Code(), 0)
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
y, _ = bench.make_frames(dec, 1 << 17, seed, snr_db=snr)
res = dec.nms(y, 10, 0.669435)
index, count = dec.compact(res["fail"]); nf = int(count.cpu()[0])
yf = y[index[:nf].long()].contiguous()
perm, parity, _ = dec.osd_front(yf)
p = dec.osd_params(3, _lib.OSD_PB, snr_db=snr)
for rep in range(2):
    print("== snr", snr, "frames", nf, "rep", rep, file=sys.stderr, flush=True)
    dec.osd_search(yf, perm, parity, p)
    torch.cuda.synchronize()
