import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np, torch, cProfile, pstats
import bench
from short_ldpc_decoding_osd_amd import Code
from short_ldpc_decoding_osd_amd import globalmap as GL
from short_ldpc_decoding_osd_amd import ms_test
from short_ldpc_decoding_osd_amd.runtime import default_decoder
from short_ldpc_decoding_osd_amd.weights import STORED_NMS1_WEIGHT
torch.cuda.set_device(0)
code = Code(); GL.set_map('code_parameters', code); GL.set_map('num_iterations', 10); GL.set_map('selected_decoder_type', 'NMS-1')
model = ms_test.Decoding_model(); model.set_check_weight(STORED_NMS1_WEIGHT)
dec = default_decoder(code)
for B in (1000, 131072):
    y_d, lab_d = bench.make_frames(dec, B, seed=20241020 + B)
    inputs = y_d.cpu().numpy(); labels = dec.unpack_bits(lab_d).cpu().numpy()
    for _ in range(3): model(inputs, labels)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    t0 = time.perf_counter()
    n = 20 if B == 1000 else 5
    for _ in range(n): model(inputs, labels)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    pr.disable()
    print(f"B={B}: {dt*1e3:.3f} ms per call")
    pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
