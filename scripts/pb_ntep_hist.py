"""TEP-count distribution of PB-OSD (order 3) on the NMS failures of one 131 072-frame batch: python scripts/pb_ntep_hist.py [snr ...]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch
from short_ldpc_decoding_osd_amd import Code, _lib
from short_ldpc_decoding_osd_amd.runtime import Decoder
import bench
dec = Decoder(Code(), 0)
for snr in [float(a) for a in sys.argv[1:]] or [2.5]:
    y, _ = bench.make_frames(dec, 1 << 17, 1, snr_db=snr)
    res = dec.nms(y, 10, 0.669435)
    index, count = dec.compact(res["fail"]); nf = int(count.cpu()[0])
    out = dec.osd_decode(y, 3, params=dec.osd_params(3, _lib.OSD_PB, snr_db=snr), index=index[:nf].contiguous())
    torch.cuda.synchronize()
    nt = out["ntep"].cpu().numpy()
    print(f"snr {snr}: OSD frames {nf}, mean TEPs {nt.mean():.1f}; frames with more than", {t: int((nt > t).sum()) for t in (64, 320, 1024, 2048, 4096, 8192, 16384, 32768)},
          "complete scans", int((nt >= 43745).sum()), flush=True)
