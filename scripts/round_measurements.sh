#!/bin/bash
# the bench / sweep lines a round's profiles/rNN quotes.  usage: scripts/round_measurements.sh <tag>   (GPU box)
TAG=${1:-m}; O=gpurun_out; mkdir -p $O
b() { local name=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/${TAG}_$name.json 2> $O/${TAG}_$name.err; echo "[$name] rc=$?"; }
b driver --gpus 1 --steps 20 --warmup 5
b default
b nms10 --workload nms10
b osd0 --workload nms10_osd0
b fs2 --workload nms10_fs2
b pb3 --workload nms10_pb3
b pb3_snr1.0 --workload nms10_pb3 --snr 1.0 --steps 40 --warmup 4
timeout -k 10 500 python scripts/snr_sweep.py --snr 1.0 3.5 6 --frames 1048576 --osd pb --order 3 --cpu-check --cpu-seconds 8 > $O/${TAG}_sweep_pb3.jsonl 2> $O/${TAG}_sweep_pb3.err; echo "[sweep pb3] rc=$?"
timeout -k 10 300 python scripts/snr_sweep.py --snr 1.0 3.5 6 --frames 1048576 --osd conv --order 2 > $O/${TAG}_sweep_conv2.jsonl 2> $O/${TAG}_sweep_conv2.err; echo "[sweep conv2] rc=$?"
timeout -k 10 300 python scripts/snr_sweep.py --snr 2.0 3.0 3 --frames 4194304 --osd pb --order 3 --stop-errors 100 > $O/${TAG}_sweep_pb3_stop100.jsonl 2> $O/${TAG}_sweep_pb3_stop100.err; echo "[sweep stop] rc=$?"
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/${TAG}_*.json")):
    try: d = json.load(open(f))
    except Exception as e: print(f, "unreadable", e); continue
    print(f.split("/")[-1], "%.4g frames/s" % d["value"], "%.4f ms/step" % d["ms_per_step"], {k: round(v, 4) for k, v in d["roofline"]["all_kernels_ms"].items()}, "frac %.3f" % d["roofline"]["frac"], d.get("fer_vs_cpu", {}).get("within_5_percent"))
for f in sorted(glob.glob("$O/${TAG}_sweep*.jsonl")):
    for l in open(f):
        d = json.loads(l); print(f.split("/")[-1], d["snr_db"], "%.4g f/s" % d["frames_per_s_incl_generation"], "fer %.5f" % d.get("fer_end_to_end", -1), "teps %.1f" % d.get("mean_teps", 0), d.get("macro_batches"), (d.get("fer_vs_cpu") or {}).get("within_5_percent"))
PY
