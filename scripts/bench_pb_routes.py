"""PB-OSD order 3 on the NMS failures of four 131 072-frame batches, three routes, HIP-event timed (GPU box):
   ldpc_osd_decode (front end as its own kernel, through the workspace) / the same with the front end inside the first PB kernel
   (reserved bit 0) /
   ldpc_osd_front + ldpc_osd_search through caller buffers.      python scripts/bench_pb_routes.py [snr ...]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import torch
from short_ldpc_decoding_osd_amd import Code, _lib
from short_ldpc_decoding_osd_amd.runtime import Decoder
import bench
dec = Decoder(Code(), 0)
for snr in [float(a) for a in sys.argv[1:]] or [2.5]:
    batches = []
    for i in range(4):
        y, _ = bench.make_frames(dec, 1 << 17, 20241020 + 1000 * i, snr_db=snr)
        res = dec.nms(y, 10, 0.669435)
        index, count = dec.compact(res["fail"]); nf = int(count.cpu()[0])
        batches.append((y, index[:nf].contiguous(), None))
    p = dec.osd_params(3, _lib.OSD_PB, snr_db=snr)
    p2 = dec.osd_params(3, _lib.OSD_PB, snr_db=snr, pb_front_inside=True)
    routes = {
        "decode, front end inside": lambda y, idx, st: dec.osd_decode(y, 3, index=idx, params=p2, out=st.setdefault("a", None)),
        "decode, front end apart": lambda y, idx, st: dec.osd_decode(y, 3, index=idx, params=p, out=st.setdefault("b", None)),
        "front + search": lambda y, idx, st: dec.osd_search(y, st["f"][0], st["f"][1], p, index=idx, out=st.setdefault("c", None)),
    }
    REPS = 12
    for name, fn in routes.items():
        states = [dict() for _ in batches]
        for (y, idx, _), st in zip(batches, states):
            k = {"decode, front end inside": "a", "decode, front end apart": "b", "front + search": "c"}[name]
            if name == "front + search":
                st["f"] = dec.osd_front(y, index=idx)
            st[k] = fn(y, idx, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REPS):
            for (y, idx, _), st in zip(batches, states):
                if name == "front + search":
                    st["f"] = dec.osd_front(y, index=idx, out=st["f"])
                fn(y, idx, st)
        e1.record(); torch.cuda.synchronize()
        print(f"snr {snr} {name:28s} {e0.elapsed_time(e1) / (4 * REPS):8.4f} ms per call (front end + PB search, 4 batches x {REPS})", flush=True)
