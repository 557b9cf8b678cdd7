import os, sys, json
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch
from short_ldpc_decoding_osd_amd import Code, _lib
from short_ldpc_decoding_osd_amd.runtime import Decoder
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from bench_pb_paths import frames
dec = Decoder(Code(), 0)
y = frames(dec, 1 << 17, 1)
res = dec.nms(y, 10, 0.669435)
index, count = dec.compact(res["fail"]); nf = int(count.cpu()[0])
yf = y[index[:nf].long()].contiguous()
perm, parity, _ = dec.osd_front(yf)
for name, F, snr, path in [("typical", nf, 2.5, None), ("fullscan256", 256, -5.0, "block")]:
    p = dec.osd_params(3, _lib.OSD_PB, snr_db=snr, pb_path=path)
    for rep in range(2):
        print("==", name, rep, file=sys.stderr, flush=True)
        dec.osd_search(yf[:F], perm[:F], parity[:F], p)
        torch.cuda.synchronize()
