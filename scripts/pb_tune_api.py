"""PB-OSD kernel time under tuning settings (ldpc_ctx_set_pb_tuning), serial launches, HIP-event timed, mean over four batches
(where a search is handed to the workgroup kernel depends on timing, so single launches scatter by +-15 %).
    python scripts/pb_tune_api.py <snr> "t2=336" "t2=352,budget_xl=32768" ...      (GPU box)"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import torch
from short_ldpc_decoding_osd_amd import Code, _lib
from short_ldpc_decoding_osd_amd.runtime import Decoder
import bench
snr = float(sys.argv[1])
dec = Decoder(Code(), 0)
p = dec.osd_params(3, _lib.OSD_PB, snr_db=snr)
batches = []
for i in range(4):
    y, _ = bench.make_frames(dec, 1 << 17, 20241020 + 1000 * i, snr_db=snr)
    res = dec.nms(y, 10, 0.669435)
    index, count = dec.compact(res["fail"]); nf = int(count.cpu()[0])
    yf = y[index[:nf].long()].contiguous()
    perm, parity, _ = dec.osd_front(yf)
    batches.append((yf, perm, parity, dec.osd_search(yf, perm, parity, p)))
torch.cuda.synchronize()
REPS = 12
for spec in ["default"] + sys.argv[2:] + ["default"]:
    dec.set_pb_tuning()
    if spec != "default":
        dec.set_pb_tuning(**{k: int(v) for k, v in (kv.split("=") for kv in spec.split(","))})
    for yf, perm, parity, out in batches:
        dec.osd_search(yf, perm, parity, p, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        for yf, perm, parity, out in batches:
            dec.osd_search(yf, perm, parity, p, out=out)
    e1.record(); torch.cuda.synchronize()
    print(f"snr {snr} {spec:40s} {e0.elapsed_time(e1) / (4 * REPS):8.4f} ms per search call (4 batches x {REPS})", flush=True)
