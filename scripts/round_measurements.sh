#!/bin/bash
# the bench / sweep lines a round's profiles/rNN quotes.  usage: scripts/round_measurements.sh <round, e.g. r04>   (GPU box)
RND=${1:-r04}; O=gpurun_out; P=$O/$RND; mkdir -p $O $P     # (copy gpurun_out/<round>/* into profiles/<round>/ afterwards)
b() { local name=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/${RND}_$name.json 2> $O/${RND}_$name.err; local rc=$?; echo "[$name] rc=$rc"; [ $rc -eq 0 ] && cp $O/${RND}_$name.json $P/bench_${name}_${RND}.json; }
b driver --gpus 1 --steps 20 --warmup 5
b default
b chain --steps 20 --warmup 5 --graph-branches 1
b nms10 --workload nms10
b osd0 --workload nms10_osd0
b fs2 --workload nms10_fs2
b pb3 --workload nms10_pb3
b pb3_snr1.0 --workload nms10_pb3 --snr 1.0 --steps 40 --warmup 4
b surface --workload surface_nms --steps 6 --warmup 2
timeout -k 10 500 python scripts/snr_sweep.py --snr 1.0 3.5 6 --frames 1048576 --osd pb --order 3 --cpu-check --cpu-seconds 8 > $P/snr_sweep_pb3_1M.jsonl 2> $O/${RND}_sweep_pb3.err; echo "[sweep pb3] rc=$?"
# (the same sweep without the CPU check: between two points the host then does nothing for 8 s with 16 threads, and the points that
#  follow a check read 5-15 % lower -- timing from this file, FER judgement from the one above)
timeout -k 10 300 python scripts/snr_sweep.py --snr 1.0 3.5 6 --frames 1048576 --osd pb --order 3 > $P/snr_sweep_pb3_1M_timing.jsonl 2> $O/${RND}_sweep_pb3_timing.err; echo "[sweep pb3 timing] rc=$?"
# (and with the point's frames generated before its timed region: inputs resident, as bench.py's contract has it)
timeout -k 10 300 python scripts/snr_sweep.py --snr 1.0 3.5 6 --frames 1048576 --osd pb --order 3 --resident > $P/snr_sweep_pb3_1M_resident.jsonl 2> $O/${RND}_sweep_pb3_resident.err; echo "[sweep pb3 resident] rc=$?"
timeout -k 10 300 python scripts/snr_sweep.py --snr 1.0 3.5 6 --frames 1048576 --osd conv --order 2 > $P/snr_sweep_conv2_1M.jsonl 2> $O/${RND}_sweep_conv2.err; echo "[sweep conv2] rc=$?"
timeout -k 10 300 python scripts/snr_sweep.py --snr 2.0 3.0 3 --frames 4194304 --batch 16384 --osd pb --order 3 --stop-errors 100 > $P/snr_sweep_pb3_stop100.jsonl 2> $O/${RND}_sweep_pb3_stop100.err; echo "[sweep stop] rc=$?"
python - <<PY
import json, glob
for f in sorted(glob.glob("$P/bench_*_${RND}.json")):
    try: d = json.load(open(f))
    except Exception as e: print(f, "unreadable", e); continue
    r = d.get("roofline", {})
    print(f.split("/")[-1], "%.4g frames/s" % d["value"], "%.4f ms/step" % d["ms_per_step"], {k: round(v, 4) for k, v in r.get("all_kernels_ms", {}).items()}, "frac %.3f" % r.get("frac", 0), d.get("fer_vs_cpu", {}).get("within_5_percent"))
for f in sorted(glob.glob("$P/snr_sweep*.jsonl")):
    for l in open(f):
        d = json.loads(l); print(f.split("/")[-1], d["snr_db"], "%.4g f/s" % d.get("frames_per_s_incl_generation", d.get("frames_per_s_resident_inputs", 0)), "fer %.5f" % d.get("fer_end_to_end", -1), "teps %.1f" % d.get("mean_teps", 0), d.get("macro_batches"), (d.get("fer_vs_cpu") or {}).get("within_5_percent"))
PY
