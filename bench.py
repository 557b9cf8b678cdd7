#!/usr/bin/env python3
"""Headline benchmark: decoded frames/s (+FER) for the (128,64) CCSDS LDPC code,
NMS-10 (+ OSD-p on the syndrome-failed frames) at Eb/N0 = 2.5 dB on synthetic AWGN frames.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload nms10_osd2|nms10_osd0|nms10|nms10_fs2|nms10_pb3|surface_nms]

One "step" = one pass of the hot path over one device-resident batch of frames:
NMS-10 -> error counters + failed-frame compaction -> OSD-p on the failures -> OSD counters, all on the GPU
with no host round trip, enqueued by ONE call into the C ABI (ldpc_pipeline_run).  The timed loop ROTATES over
--batches (default 4) distinct pre-generated batches, each with its own buffers, so that inputs + outputs of one
rotation (4 x 136 MB) exceed the 256 MiB Infinity Cache and every step streams from / to HBM.

Ranks: one process per GPU (torch.distributed, backend nccl = RCCL).  Under a launcher (torchrun: WORLD_SIZE set)
this process is one rank.  Started plainly with --gpus N > 1, it spawns the N ranks itself as child processes
BEFORE touching any GPU, relays rank 0's JSON line and fails if any rank fails or fewer than N GPUs are visible.
Frames shard across ranks (weak scaling: fixed per-GPU batch); the only collective is ONE all-reduce of the
error counters per measurement, inside the timed region.

Prints ONE JSON line (rank 0) with the contract fields plus "roofline" (dominant kernel, HIP-event timed),
"cpu_baseline" (the C oracle = scalar port of the same math on this host's cores, bounded sample, N = 1 only;
with its FER and, as `reference_equivalent`, the dense NumPy mirror of the reference's formulation on one core)
and "fer_vs_cpu" (north_star acceptance: FER within +-5 % of the float CPU path in the same run).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SNR_DB = 2.5
T_ITERS = 10
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
NMS_BYTES_PER_FRAME = 1040  # 512 B LLR in + 512 B posterior out + 16 B hard word
OSD_BYTES_PER_FRAME = 536   # 512 B channel values in + 16 B codeword + 4 B metric + 4 B TEP id

WORKLOADS = {
    # name: (osd order or None, default per-GPU batch, BASELINE.json config it is)
    "nms10": (None, 65536, "configs[1]: NMS-10, 65 536 frames, 1 GPU"),
    "nms10_osd0": (0, 65536, "configs[2]: NMS-10 + OSD-0, 65 536 frames"),
    "nms10_osd2": (2, 131072, "configs[3]: NMS-10 + OSD-2, 2^20 frames over 8 GPUs = 131 072 per GPU"),
    # not headline lines: the other two searches at 2.5 dB, for kernel timing (configs[4] sweeps them over SNR,
    # scripts/snr_sweep.py)
    "nms10_fs2": (2, 131072, "NMS-10 + FS-OSD order 2 (beta 0.1, tau_e 6.5, tau_psc 30)"),
    "nms10_pb3": (3, 131072, "NMS-10 + PB-OSD order 3 (configs[4] at one SNR point)"),
    # the reference's own call surface of the NMS test stage: Decoding_model.__call__(inputs, labels) with NumPy in and Python
    # lists out, exactly the body of ldpc_128_testing.py:117-131 (not a headline line: host conversions and PCIe are inside)
    "surface_nms": (None, 131072, "stage-5 call surface: ms_test.Decoding_model(inputs, labels), NumPy in, lists of rows out"),
}
OSD_ALGO = {"nms10_fs2": 1, "nms10_pb3": 2}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=os.environ.get("LDPC_BENCH_WORKLOAD", "nms10_osd2"), choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="frames per GPU and step (0 = the workload's default)")
    ap.add_argument("--batches", type=int, default=4, help="distinct pre-generated batches the timed loop rotates over")
    ap.add_argument("--no-graph", dest="graph", action="store_false",
                    help="time eager ldpc_pipeline_run calls instead of replaying the steps from a captured HIP graph "
                         "(one graph = one rotation over the distinct batches; the eager form pays ~18 us of launch / ctypes "
                         "time per 0.3 ms step)")
    ap.add_argument("--graph-branches", type=int, default=4,
                    help="parallel branches of the captured rotation: the distinct batches are independent, so batch i is captured on "
                         "stream i mod N and the branches join at the end of the graph -- the ~5 us dispatch gap after every kernel "
                         "and the kernel tails of one batch then hide behind the kernels of another (1 = one chain)")
    ap.add_argument("--event-steps", type=int, default=24,
                    help="steps of the SEPARATE pass, after the timed region, whose hot kernels are bracketed by HIP events "
                         "(the roofline's live kernel durations; no event is recorded inside the timed region)")
    ap.add_argument("--rank-timeout", type=float, default=900.0, help="--gpus N > 1 without a launcher: seconds the ranks may take")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap-pass", dest="overlap_pass", action="store_false",
                    help="skip the informative 3-stream pass that follows the timed region")
    ap.add_argument("--snr", type=float, default=SNR_DB,
                    help="Eb/N0 in dB of the synthetic frames (the headline metric is quoted at 2.5 dB; other values are for "
                         "kernel timing of the configs[4] sweep points, e.g. --workload nms10_pb3 --snr 1.0)")
    ap.add_argument("--osd-route", choices=["decode", "front+search"], default="front+search",
                    help="how the OSD stage of a step is issued: 'decode' = ldpc_osd_decode inside ldpc_pipeline_run (conventional order 2: "
                         "ONE fused front-end + scan kernel, nothing written to a workspace; other searches: front end into the "
                         "stream's workspace, then the search); 'front+search' = the two kernels through caller buffers, timed apart")
    ap.add_argument("--pb-tuning", default="",
                    help="PB-OSD tuning of the context, e.g. 'late_div=1,t2=512' (ldpc_ctx_set_pb_tuning; changes no result, "
                         "only how searches are cut into chunks and handed on); recorded in the line's config")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams of the timed region; >1 keeps several batches in flight (batch i runs on stream i mod N)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------
# rank launcher: the parent never touches a GPU
# ---------------------------------------------------------------------------------------------------------
def visible_gpus():
    """Number of GPUs this process may use, WITHOUT initialising a HIP runtime in it (the parent of the ranks must stay
    GPU-free): the visibility variables if set, else a short-lived child process that asks torch."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    try:
        out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=300)
        return int(out.stdout.strip().splitlines()[-1])
    except Exception:   # noqa: BLE001
        return 0


def spawn_ranks(args, argv, worker=None, have=None, poll=0.2):
    """Start the N ranks as child processes and relay rank 0's JSON line.  Every child is polled: the first non-zero exit
    (or the overall timeout) kills the others -- a rank that dies after init_process_group would otherwise leave rank 0
    waiting in an RCCL barrier for ever, and the parent with it.  `worker` (argv prefix of a rank; default:
    this file) and `have` exist for the CPU unit test of this function."""
    import socket
    n = args.gpus
    have = visible_gpus() if have is None else have
    if have < n:
        raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible -- refusing to report a {n}-GPU number")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    worker = worker or [sys.executable, os.path.abspath(__file__)]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen(worker + argv, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading
    out = []
    reader = threading.Thread(target=lambda: out.append(procs[0].stdout.read()), daemon=True)   # (drain rank 0's pipe while polling)
    reader.start()
    deadline = time.monotonic() + float(getattr(args, "rank_timeout", 900.0))
    failed = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = f"rank {bad[0][0]} exited with code {bad[0][1]}"
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() > deadline:
            failed = f"ranks still running after {getattr(args, 'rank_timeout', 900.0):.0f} s"
            break
        time.sleep(poll)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
        raise SystemExit(f"bench.py --gpus {n}: {failed}; the other ranks were killed")
    reader.join(timeout=10)
    line = [ln for ln in (out[0].decode() if out else "").splitlines() if ln.startswith("{")]
    if not line:
        raise SystemExit(f"bench.py --gpus {n}: no JSON line from rank 0")
    print(line[-1], flush=True)


# ---------------------------------------------------------------------------------------------------------
def make_frames(dec, B, seed, snr_db=SNR_DB):
    """Synthetic test frames on the device (Testing_data_gen_128/data_generating.py:13-51):
    random message . G, BPSK 0 -> +1, y = (1 - 2c)(1 + sigma N(0,1)), unscaled float32."""
    import numpy as np
    import torch
    g = torch.Generator(device=dec.device).manual_seed(seed)
    G = torch.from_numpy(dec.code.G).to(device=dec.device, dtype=torch.float32)
    sigma = float(np.sqrt(1.0 / (2.0 * (dec.k / dec.n) * 10.0 ** (snr_db / 10.0))))
    y = torch.empty((B, dec.n), dtype=torch.float32, device=dec.device)
    labels = torch.empty((B, dec.words), dtype=torch.int64, device=dec.device)
    chunk = 1 << 16
    for s in range(0, B, chunk):
        e = min(B, s + chunk)
        msg = torch.randint(0, 2, (e - s, dec.k), device=dec.device, generator=g).to(torch.float32)
        cw = (msg @ G).remainder_(2)
        noise = torch.randn((e - s, dec.n), device=dec.device, generator=g)
        y[s:e] = (1 - 2 * cw) * (1 + sigma * noise)
        labels[s:e] = dec.pack_bits(cw.to(torch.uint8))
    return y, labels


def pmc_profile(kernel, workload, B):
    """(HBM bytes per launch, VALU issue ratio, file) of `kernel` from the committed rocprofv3 PMC passes
    (scripts/profile_bench.sh + scripts/pmc_summary.py: FETCH_SIZE and WRITE_SIZE in separate runs, FETCH_SIZE
    doubled as the gfx950 guide prescribes) -- only when the profile was taken on this very workload; else nulls.
    The PB-OSD entry of the bench line is three kernels: their traffic is summed, the ratio is the chunk kernel's."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_counters_*.json")), reverse=True):
        try:
            prof = json.load(open(path))
        except (OSError, ValueError):
            continue
        if prof.get("bench_workload") != workload or prof.get("frames_per_launch") != B:
            continue
        rows = prof.get("per_launch_mean", {})
        if kernel.startswith("pb_osd"):
            sel = [c for n, c in rows.items() if "::pb_" in n and "clear" not in n and "FETCH_SIZE" in c and "WRITE_SIZE" in c]
            if sel:
                main = [c for n, c in rows.items() if "pb_wave_kernel" in n]
                return (sum((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 for c in sel),
                        main[0].get("valu_issue_frac") if main else None, os.path.relpath(path, ROOT))
            continue
        for name, c in rows.items():
            if kernel in name and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                issue = c.get("valu_issue_frac")
                if issue is None and c.get("GRBM_GUI_ACTIVE") and "SQ_ACTIVE_INST_VALU" in c:
                    issue = c["SQ_ACTIVE_INST_VALU"] / (c["GRBM_GUI_ACTIVE"] / 8.0 / 4.0 * 1024.0)
                return (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0, issue, os.path.relpath(path, ROOT)
    return None, None, None


def cpu_counts(code_G, code_H, y, cw, alpha, order, algo=0, snr_db=SNR_DB, iters=T_ITERS):
    """One batch through the CPU port (oracle/ldpc_oracle.c): NMS, then the OSD of the workload on the syndrome failures.
    int64[5] = {frames, NMS frame errors, undetected NMS errors, syndrome failures, OSD failures}.  The checker, timed
    beside the GPU path -- never part of it (only bench.py's cpu_baseline leg, scripts/snr_sweep.py --cpu-check and the
    tests may call into oracle/)."""
    import numpy as np

    from oracle import c_oracle
    soft = c_oracle.nms(code_H, y, iters, alpha)
    _, fail, cnt = c_oracle.evaluate(code_H, soft, cw)
    nfail = int(cnt["synd_fail"])
    wrong = nfail
    if order is not None:
        idx = np.flatnonzero(fail)
        wrong = 0
        if idx.size:
            if algo == 2:
                ok = c_oracle.pb_osd(code_G, y[idx], cw[idx], order, snr_db)["correct"]
            elif algo == 1:
                ok = c_oracle.fs_osd(code_G, y[idx], cw[idx], order)["correct_ref"]
            else:
                ok = c_oracle.conv_osd(code_G, y[idx], cw[idx], order)["correct"]
            wrong = int((~ok).sum())
    return np.array([y.shape[0], cnt["frame_err"], cnt["undetected"], nfail, wrong], dtype=np.int64)


def host_cores():
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return max(1, min(cores, 16))   # a one-GPU box shares its host: 16 cores per GPU


def cpu_baseline(code_G, code_H, order, alpha, algo=0, snr_db=SNR_DB, seconds_target=12.0):
    """The C oracle (scalar port of the same math) on this host: frames are split over the host cores this
    process may use (the C calls release the GIL), bounded sample; the one-thread rate and the sample's FER are
    reported alongside, plus the dense NumPy mirror of the reference's own formulation on one core."""
    import concurrent.futures

    import numpy as np

    from oracle import np_oracle
    rng = np.random.default_rng(20241020)
    cores = host_cores()
    osd_name = {0: "OSD", 1: "FS-OSD", 2: "PB-OSD"}[algo]

    def once(yb, cb):
        return cpu_counts(code_G, code_H, yb, cb, alpha, order, algo, snr_db)

    frames = 2000 if algo == 0 else 300        # (the port's PB-OSD replays the frontier list literally: ~1 ms per long search)
    y, cw = np_oracle.make_frames(code_G, snr_db, frames, rng)
    t0 = time.perf_counter()
    once(y, cw)
    rate1 = frames / (time.perf_counter() - t0)                        # one thread, also the warm-up
    per = int(max(200, min(100000, rate1 * seconds_target)))             # frames per worker
    y, cw = np_oracle.make_frames(code_G, snr_db, per * cores, rng)
    parts = [(y[i * per:(i + 1) * per], cw[i * per:(i + 1) * per]) for i in range(cores)]
    with concurrent.futures.ThreadPoolExecutor(max_workers=cores) as ex:
        t0 = time.perf_counter()
        c = sum(ex.map(lambda p: once(*p), parts))
        dt = time.perf_counter() - t0
    # "reference-equivalent CPU" (SURVEY 8(d)(i)): the reference's dense [B,64,128] formulation (ms_test.py:124-228) and
    # its per-frame int-matmul OSD (convention_osd.py:49-76), restated in NumPy, one core (conventional OSD only)
    ref = None
    if algo == 0:
        nref = 400
        yr, cr = y[:nref], cw[:nref]
        t0 = time.perf_counter()
        soft = np_oracle.nms_dense(yr, code_H, T_ITERS, alpha)[-1]
        t_nms = time.perf_counter() - t0
        _, _, _, failed = np_oracle.evaluate(soft, cr, code_H)
        t0 = time.perf_counter()
        nosd = 0
        if order is not None:
            teps = np_oracle.tep_matrix(64, order)
            for i in list(failed)[:40]:
                yp, lp, Gp, _, _ = np_oracle.swapped_info(yr[i], cr[i], code_G)
                np_oracle.convention_osd(yp, lp, Gp, order, teps)
                nosd += 1
        t_osd = (time.perf_counter() - t0) / max(nosd, 1) * len(failed)      # scaled to all failures of the sample
        ref_rate = nref / (t_nms + (t_osd if order is not None else 0.0))
        ref = dict(value=ref_rate, unit="frames/s", cores=1, kind="port",
                   sample=f"dense NumPy mirror of ms_test.py:124-228 on {nref} frames ({nref / t_nms:.0f} frames/s)"
                          + (f" + NumPy convention_osd_main order {order} timed on {nosd} of the {len(failed)} "
                             f"failures ({t_osd / max(len(failed), 1) * 1e3:.0f} ms per failed frame)" if order is not None else ""))
    e2e = int(c[4] + c[2]) if order is not None else int(c[1])
    out = dict(value=per * cores / dt, unit="frames/s", cores=cores, kind="port",
               sample=f"{per * cores} frames at {snr_db} dB through oracle/ldpc_oracle.c (gcc -O2, scalar code, {cores} "
                      f"threads of {per} frames): NMS-{T_ITERS}"
                      + (f" + {osd_name}-{order} on the syndrome failures" if order is not None else "")
                      + f", {dt:.1f} s; one thread alone: {rate1:.0f} frames/s",
               frames=int(c[0]), fer_nms=float(c[1] / c[0]), syndrome_fail_rate=float(c[3] / c[0]),
               fer_end_to_end=float(e2e / c[0]), frame_errors_end_to_end=e2e)
    if ref is not None:
        out["reference_equivalent"] = ref
    return out


def run_surface(args):
    """--workload surface_nms: the reference's stage-5 batch body through the package's mirror of its call surface
    (short_ldpc_decoding_osd_amd/ms_test.py = LDPC_128/Ldpc_128_testing/ms_test.py:26-70): NumPy `inputs` [B,128] f32 and
    `labels` [B,128] int in, (fer, ber, undetected, (buffer_inputs, buffer_labels)) out, the buffers being Python lists of
    T + 1 rows per failed frame.  Timed wall-clock around the call (host conversions, both PCIe directions and the list
    building are the surface), at the reference's batch (1000, ldpc_128_testing.py:20) and at --batch (131 072); the device
    part alone and the trajectory kernel are timed with events beside it."""
    import numpy as np
    import torch

    from short_ldpc_decoding_osd_amd import Code
    from short_ldpc_decoding_osd_amd import globalmap as GL
    from short_ldpc_decoding_osd_amd import ms_test
    from short_ldpc_decoding_osd_amd.runtime import default_decoder
    from short_ldpc_decoding_osd_amd.weights import STORED_NMS1_WEIGHT, softplus32

    torch.cuda.set_device(0)
    code = Code()
    GL.set_map('code_parameters', code)
    GL.set_map('num_iterations', T_ITERS)
    GL.set_map('selected_decoder_type', 'NMS-1')
    model = ms_test.Decoding_model()
    model.set_check_weight(STORED_NMS1_WEIGHT)
    dec = default_decoder(code)
    alpha = float(softplus32(STORED_NMS1_WEIGHT))
    big = args.batch or WORKLOADS["surface_nms"][1]
    steps = max(2, min(args.steps, 20))
    lines = {}
    for B in (1000, big):
        y_d, lab_d = make_frames(dec, B, seed=20241020 + B, snr_db=args.snr)
        inputs = y_d.cpu().numpy()
        labels = dec.unpack_bits(lab_d).cpu().numpy()                    # [B,128] int64, one integer per bit as the reference holds them
        for _ in range(max(1, min(args.warmup, 3))):
            model(inputs, labels)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fer, ber, und, (bi, bl) = model(inputs, labels)
        torch.cuda.synchronize()
        call_ms = 1e3 * (time.perf_counter() - t0) / steps
        nfail = len(model.last_failed_index)
        # the device part alone (inputs resident): NMS, counters, failure list, rows of the failures
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        lab_bits = dec.pack_bits(torch.from_numpy(labels).to(dec.device))
        dev_ms, rows_ms = [], []
        for _ in range(steps):
            e[0].record()
            res = dec.nms(y_d, T_ITERS, alpha, want_traj=False)
            dec.eval_counts(res["hard"], lab_bits, res["fail"])
            index, count = dec.compact(res["fail"])
            e[1].record()
            dec.nms_traj_rows(y_d, index, count, max(nfail, 1), T_ITERS, alpha)
            e[2].record()
            torch.cuda.synchronize()
            dev_ms.append(e[0].elapsed_time(e[2])); rows_ms.append(e[1].elapsed_time(e[2]))
        e[0].record()
        for _ in range(steps):
            dec.nms(y_d, T_ITERS, alpha, want_traj=True, want_soft=False, want_hard=False, want_fail=False)      # round 3's surface: [T][B][n] for every frame
        e[3].record(); torch.cuda.synchronize()
        full_traj_ms = e[0].elapsed_time(e[3]) / steps
        row_bytes = nfail * (T_ITERS + 1) * 512
        lines[B] = dict(frames=B, frames_per_s=B / (call_ms * 1e-3), call_ms=call_ms, device_ms=float(np.median(dev_ms)),
                        traj_rows_kernel_ms=float(np.median(rows_ms)), failed_frames=nfail, fer=fer, ber=float(ber), undetected=und,
                        buffer_rows=len(bi), trajectory_bytes_per_input_frame=row_bytes / B,
                        full_trajectory_bytes_per_input_frame=T_ITERS * 512, full_trajectory_kernel_ms=full_traj_ms)
    b = lines[big]
    rows_ms = b["traj_rows_kernel_ms"]
    nbytes = b["failed_frames"] * (512 + (T_ITERS + 1) * 512)          # channel row in, T + 1 rows out per failed frame
    res = {
        "metric": "decoded frames/sec through the reference call surface Decoding_model(inputs, labels), NMS-10 @ 2.5 dB",
        "timed_region": "wall clock around model(inputs, labels): host conversion + H2D + kernels + D2H of the failed frames' rows + list building",
        "value": b["frames_per_s"], "unit": "frames/s", "n_gpus": 1, "steps": steps, "warmup": args.warmup, "ms_per_step": b["call_ms"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"surface_nms -- {WORKLOADS['surface_nms'][2]}", "code": "CCSDS (128,64)", "snr_db": args.snr,
                   "nms_iterations": T_ITERS, "alpha": alpha, "frames_per_call": big,
                   "note": "NOT the headline metric: host buffers cross PCIe in both directions inside the timed region"},
        "surface": {"reference_batch_1000": lines[1000], f"batch_{big}": b},
        "roofline": {"bound": "hbm", "kernel": "nms_qc16_kernel<ROWS> (ldpc_nms_traj_rows)", "achieved": nbytes / (rows_ms * 1e-3) / 1e9,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / (rows_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                     "avg_launch_ms": rows_ms, "algorithmic_bytes_per_launch": float(nbytes),
                     "note": "the listed frames decoded again with their T + 1 rows written in the reference's buffer order; issue-bound like the decoder"},
    }
    if not args.no_cpu_baseline:
        from oracle import np_oracle
        y_np = y_d[:1000].cpu().numpy() if big >= 1000 else inputs
        cw_np = dec.unpack_bits(lab_d[:1000].contiguous()).cpu().numpy()
        t0 = time.perf_counter()
        outs = np_oracle.nms_dense(y_np, code.H, T_ITERS, alpha)
        _, _, _, idx = np_oracle.evaluate(outs[-1], cw_np, code.H)
        np_oracle.collect_failed(outs, cw_np, idx)
        dt = time.perf_counter() - t0
        res["cpu_baseline"] = dict(value=y_np.shape[0] / dt, unit="frames/s", cores=1, kind="port",
                                   sample=f"dense NumPy mirror of ms_test.py:99-228 + get_eval + collect_failed_output_selective on {y_np.shape[0]} frames, {dt:.1f} s")
    print(json.dumps(res), flush=True)


def run_rank(args):
    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("LDPC_BENCH_FORCE_DIST"):   # (the variable rehearses the RCCL path with one rank)
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from short_ldpc_decoding_osd_amd import Code
    from short_ldpc_decoding_osd_amd._lib import TIMING_SLOTS
    from short_ldpc_decoding_osd_amd.pipeline import BatchPipeline
    from short_ldpc_decoding_osd_amd.runtime import Decoder
    from short_ldpc_decoding_osd_amd.sharding import allreduce_counters
    from short_ldpc_decoding_osd_amd.weights import STORED_NMS1_WEIGHT, softplus32

    alpha = float(softplus32(STORED_NMS1_WEIGHT))      # the reference's shipped (untrained) NMS-1 weight, ms_test.py:73
    order, default_batch, cfg_name = WORKLOADS[args.workload]
    B = args.batch or default_batch
    nb = max(1, args.batches, args.streams)
    dec = Decoder(Code(), local_rank)
    if args.pb_tuning:
        dec.set_pb_tuning(**{k: int(v) for k, v in (kv.split("=") for kv in args.pb_tuning.split(","))})
    algo = OSD_ALGO.get(args.workload, 0)
    pipes = []
    for i in range(nb):                                # distinct batches, each with its own buffers
        y, labels = make_frames(dec, B, seed=20241020 + rank + 1000 * i, snr_db=args.snr)
        pipes.append(BatchPipeline(dec, B, T_ITERS, alpha, osd_order=order, osd_algo=algo, snr_db=args.snr,
                                   keep_front=args.osd_route == "front+search").bind(y, labels))
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(1, max(1, args.streams))]

    def run_step(k, slot=-1):
        with torch.cuda.stream(streams[(k % nb) % len(streams)]):
            pipes[k % nb].run(timing_slot=slot)

    for k in range(max(args.warmup, nb)):
        run_step(k)
    torch.cuda.synchronize()
    # The timed region replays the steps from a captured HIP graph: one graph = one rotation over the nb distinct batches
    # (nb x the launches of ldpc_pipeline_run), so a 0.3 ms step no longer carries ~18 us of launch and ctypes time per
    # step (the driver-timed step was 6 % above the kernel sum).  Steps beyond whole rotations, several
    # streams, or a failed capture run eagerly; the JSON line says which.
    graph, timed_region = None, "eager ldpc_pipeline_run calls"
    graph_info = {"graph_steps_per_launch": 0, "graph_branches": 0, "overlapped": len(streams) > 1}
    rpg = 1
    if args.graph and len(streams) == 1 and args.steps >= nb:
        try:
            side = torch.cuda.Stream()
            nbr = max(1, min(args.graph_branches, nb))
            branch = [side] + [torch.cuda.Stream() for _ in range(1, nbr)]

            # rotations per graph launch: a graph ends with a join of its branches, and the tail of the slowest branch of one
            # launch does not overlap the head of the next -- with the driver's 20 steps that was five joins in 5.8 ms.  Up to
            # 8 rotations (32 steps) go into one graph; batch i stays on branch i mod N through all of them (its buffers are
            # reused step after step, so the branch's stream order is also the data dependence).
            rot_total = args.steps // nb
            rpg = max(r for r in range(1, 9) if rot_total % r == 0)

            def rotation(reps=1):
                for st in branch[1:]:
                    st.wait_stream(side)                 # fork
                for _ in range(reps):
                    for i, p in enumerate(pipes):
                        with torch.cuda.stream(branch[i % nbr]):
                            p.run()
                for st in branch[1:]:
                    side.wait_stream(st)                 # join

            with torch.cuda.stream(side):
                for st in branch:                        # every capture stream's OSD workspace (PB-OSD: lists, tables) is sized first
                    with torch.cuda.stream(st):
                        if order is not None:
                            dec.osd_reserve_stream(B, pipes[0]._p.osd)
                rotation()                               # (one eager rotation on the capture streams: nothing is allocated under capture)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                rotation(rpg)
            graph = g
            for _ in range(max(1, -(-args.warmup // (nb * rpg)))):      # the warm-up steps once more, as replays (a graph's first launch uploads it)
                g.replay()
            torch.cuda.synchronize()
            timed_region = (f"HIP graph replay, {nb * rpg} steps ({rpg} rotation(s) over the {nb} distinct batches) per graph launch, "
                            f"{nbr} parallel branch(es)")
            graph_info = {"graph_steps_per_launch": nb * rpg, "graph_branches": nbr, "overlapped": nbr > 1}
        except Exception as exc:   # noqa: BLE001
            graph = None
            timed_region = f"eager ldpc_pipeline_run calls (graph capture failed: {type(exc).__name__})"
            torch.cuda.synchronize()
    for p in pipes:
        p.reset_counters()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if graph is not None:
        for _ in range(args.steps // (nb * rpg)):
            graph.replay()
        for k in range(args.steps - args.steps % nb, args.steps):
            run_step(k)
    else:
        for k in range(args.steps):
            run_step(k)
    for st in streams[1:]:
        torch.cuda.current_stream().wait_stream(st)
    total = pipes[0].counters().clone()
    for p in pipes[1:]:
        total += p.counters()
    counters = allreduce_counters(total)               # the path's one exchange step (RCCL over xGMI)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dec.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    c = counters.cpu().numpy().astype(np.int64)

    # error counts over the DISTINCT frames only (batch i was decoded steps_i times with identical results)
    distinct = torch.zeros(8, dtype=torch.int64, device=dec.device)
    for i, p in enumerate(pipes):
        runs = len(range(i, args.steps, nb))
        if runs:
            distinct += p.counters() // runs
    d = allreduce_counters(distinct).cpu().numpy().astype(np.int64)

    # kernel durations for the roofline: a SEPARATE pass with the library's HIP events around the hot kernels (none inside
    # the timed region; recording five events costs ~17 us of a 0.3 ms step)
    nev = max(1, min(args.event_steps, TIMING_SLOTS))
    for k in range(nev):
        pipes[k % nb].run(timing_slot=k)
    torch.cuda.synchronize()
    timed_slots = list(range(nev))

    # informative second pass (not the headline): the same steps with three batches in flight on three streams,
    # where the NMS of one batch overlaps the OSD kernels (and the kernel tails) of another
    overlap = None
    if args.streams == 1 and args.overlap_pass and order is not None and nb >= 3:
        ost = [torch.cuda.current_stream(), torch.cuda.Stream(), torch.cuda.Stream()]
        for k in range(6):
            with torch.cuda.stream(ost[k % 3]):
                pipes[k % 3].run()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        t1 = time.perf_counter()
        for k in range(args.steps):
            with torch.cuda.stream(ost[k % 3]):
                pipes[k % 3].run()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        e2 = time.perf_counter() - t1
        if dist is not None:
            t = torch.tensor([e2], dtype=torch.float64, device=dec.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e2 = float(t.item())
        overlap = {"streams": 3, "value": B * args.steps * world / e2, "unit": "frames/s", "ms_per_step": 1e3 * e2 / args.steps,
                   "note": "same step, three independent batches in flight on three streams of one context "
                           "(bench.py --streams 3 makes this the timed region)"}

    frames_total = int(c[0])
    assert frames_total == B * args.steps * world, (frames_total, B, args.steps, world)
    value = frames_total / elapsed
    res = {
        # (BASELINE.json's metric, word for word, on the workload it is quoted on; the other workloads say what they decode)
        "metric": "decoded frames/sec + FER, (128,64) LDPC %s @ %s dB" % (
            {"nms10": "NMS-10", "nms10_osd0": "NMS-10+OSD-0", "nms10_osd2": "NMS-10+OSD-2", "nms10_fs2": "NMS-10+FS-OSD-2",
             "nms10_pb3": "NMS-10+PB-OSD-3"}[args.workload], ("%.2f" % args.snr).rstrip("0").rstrip(".") if args.snr != 2.5 else "2.5"),
        "timed_region": timed_region,
        # (machine-readable form of the above -- with parallel branches ms_per_step is a throughput reciprocal, not
        #  a step latency; `--graph-branches 1` times one chain)
        "timed_region_info": graph_info,
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload} -- {cfg_name}", "code": "CCSDS (128,64)", "snr_db": args.snr,
                   "nms_iterations": T_ITERS, "alpha": alpha, "osd_order": order, "frames_per_gpu": B,
                   "global_frames_per_step": B * world, "parallelism": f"frame-sharded x{world}",
                   "rccl_ranks": dist.get_world_size() if dist is not None else 1,
                   "distinct_batches_per_gpu": nb, "distinct_frames": int(d[0]),
                   "bytes_in_plus_out_per_rotation": int(nb * B * (NMS_BYTES_PER_FRAME + 1)),
                   "nms_kernel": {1: "generic", 2: "qc16"}[dec.nms_kernel], "streams": len(streams), "osd_route": args.osd_route,
                   **({"pb_tuning": dec.pb_tuning()} if algo == OSD_ALGO.get("nms10_pb3") else {})},
        "fer": {"frames": int(d[0]), "nms_frame_error_rate": d[1] / max(d[0], 1), "nms_syndrome_fail_rate": d[4] / max(d[0], 1),
                "nms_undetected": int(d[3]), "nms_ber": d[2] / max(d[0] * dec.n, 1)},
    }
    e2e_errors = int(d[1])
    if order is not None:
        osd_frames, osd_wrong, teps = int(d[5]), int(d[6]), int(d[7])
        e2e_errors = osd_wrong + int(d[3])
        res["fer"].update({"osd_frames": osd_frames, "osd_fail_rate_given_nms_fail": osd_wrong / max(osd_frames, 1),
                           "end_to_end_fer": e2e_errors / max(d[0], 1), "frame_errors_end_to_end": e2e_errors,
                           "mean_teps": teps / max(osd_frames, 1)})
    if overlap is not None:
        res["overlap"] = overlap
    fer_ok = True
    if rank == 0:
        # roofline of the dominant kernel from the HIP events recorded on the launch stream
        tm = np.array([pipes[0].timing(k) for k in timed_slots])   # (slots are per context: any pipeline reads them)
        nms_name = {1: "nms_generic_kernel", 2: "nms_qc16_kernel"}[dec.nms_kernel]
        kern = {nms_name: (float(tm[:, 0].mean()), NMS_BYTES_PER_FRAME * B)}
        if order is not None:
            f_per_step = c[5] / (args.steps * world)
            if tm[:, 1].mean() > 0.02:   # two-kernel OSD (front end + search through the workspace)
                kern["osd_front_kernel"] = (float(tm[:, 1].mean()), (512 + 640) * f_per_step)
                sname = {1: "osd_fs_kernel", 2: "pb_osd (singles + chunk + workgroup kernels)"}.get(algo, "osd_search2r_kernel" if order == 2 else "osd_search_kernel")
                kern[sname] = (float(tm[:, 2].mean()), (1152 + 24) * f_per_step)
            else:                        # ldpc_osd_decode: one duration (conventional order 2: ONE kernel, osd_fused2r_kernel;
                #                          PB-OSD: the front end inside the singles kernel; others: front end into the workspace + search)
                dname = {1: "osd_front_kernel + osd_fs_kernel (through the workspace)",
                         2: "osd_front_kernel + pb_osd (through the workspace)"}.get(
                             algo, "osd_fused2r_kernel (front end + order-2 scan)" if order == 2 else "osd_front_kernel + osd_search_kernel (through the workspace)")
                kern[dname] = (float(tm[:, 2].mean()), OSD_BYTES_PER_FRAME * f_per_step)
        name = max(kern, key=lambda k: kern[k][0])
        ms, nbytes = kern[name]
        achieved = nbytes / (ms * 1e-3) / 1e9
        traffic, issue, prof_path = pmc_profile(name if name.startswith("pb_osd") else name.split(" ")[0],
                                                  args.workload + ("_fused" if args.osd_route == "decode" and order is not None else "") + ("" if args.snr == SNR_DB else f"_snr{args.snr}"), B)
        res["roofline"] = {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "valu_issue_ratio": issue, "pmc_profile": prof_path,
                           "avg_launch_ms": ms, "timed_launches": len(timed_slots), "timed_in": "separate event-bracketed pass after the timed region",
                           "algorithmic_bytes_per_launch": float(nbytes),
                           "all_kernels_ms": {k: v[0] for k, v in kern.items()},
                           "all_kernels_GBps": {k: v[1] / (v[0] * 1e-3) / 1e9 for k, v in kern.items()},
                           "note": "no contraction on this path (no MFMA); the kernels are instruction-issue bound: valu_issue_ratio = "
                                   "SQ_ACTIVE_INST_VALU (= one 4-cycle issue slot per vector instruction, it equals SQ_INSTS_VALU) / "
                                   "the SIMD quad-cycles of the launch (GRBM_GUI_ACTIVE / 8 XCDs / 4 x 1024 SIMDs), from the committed "
                                   "PMC pass; a RATIO, not a fraction: instructions of the 2-cycle class (add, mul, fma, xor) make it "
                                   "exceed 1 when the vector ALUs never idle; HBM fraction reported as mandated -- see DESIGN.md 5"}
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(dec.code.G, dec.code.H, order, alpha, algo=algo, snr_db=args.snr)
            res["cpu_baseline"] = cpu
            # north_star acceptance: FER within +-5 % (relative) of the float CPU path of the same run, judged when both
            # sides hold >= 1600 frame errors; sigma of the ratio of two Poisson counts
            eg, ec = e2e_errors, cpu["frame_errors_end_to_end"]
            fg, fc = eg / max(d[0], 1), cpu["fer_end_to_end"]
            rel = fg / fc - 1.0 if fc > 0 else float("nan")
            two_sigma = 2.0 * float(np.sqrt(1.0 / max(eg, 1) + 1.0 / max(ec, 1)))
            enough = eg >= 1600 and ec >= 1600
            fer_ok = (abs(rel) <= 0.05) if enough else True
            res["fer_vs_cpu"] = {"gpu_fer_end_to_end": fg, "cpu_fer_end_to_end": fc, "fer_rel_diff_vs_cpu": rel,
                                 "two_sigma_of_rel_diff": two_sigma, "gpu_frame_errors": eg, "cpu_frame_errors": ec,
                                 "judged": enough, "within_5_percent": bool(fer_ok) if enough else None}
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    if not fer_ok:
        raise SystemExit("bench.py: GPU FER differs from the CPU float path by more than 5 % with >= 1600 errors on both sides")


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.workload == "surface_nms":
        if args.gpus != 1:
            raise SystemExit("bench.py --workload surface_nms is a one-GPU, one-process measurement of the reference call surface")
        run_surface(args)
    elif args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args, argv)
    else:
        run_rank(args)


if __name__ == "__main__":
    main()
