#!/bin/bash
# the three sweep files of profiles/rNN (the sweep part of scripts/round_measurements.sh).  usage: scripts/sweep_measurements.sh <tag>
TAG=${1:-m}; O=gpurun_out; mkdir -p $O
timeout -k 10 500 python scripts/snr_sweep.py --snr 1.0 3.5 6 --frames 1048576 --osd pb --order 3 --cpu-check --cpu-seconds 8 > $O/${TAG}_sweep_pb3.jsonl 2> $O/${TAG}_sweep_pb3.err; echo "[sweep pb3] rc=$?"
timeout -k 10 300 python scripts/snr_sweep.py --snr 1.0 3.5 6 --frames 1048576 --osd conv --order 2 > $O/${TAG}_sweep_conv2.jsonl 2> $O/${TAG}_sweep_conv2.err; echo "[sweep conv2] rc=$?"
timeout -k 10 300 python scripts/snr_sweep.py --snr 2.0 3.0 3 --frames 4194304 --osd pb --order 3 --stop-errors 100 > $O/${TAG}_sweep_pb3_stop100.jsonl 2> $O/${TAG}_sweep_pb3_stop100.err; echo "[sweep stop] rc=$?"
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/${TAG}_sweep*.jsonl")):
    for l in open(f):
        d = json.loads(l); print(f.split("/")[-1], d["snr_db"], "%.4g f/s" % d["frames_per_s_incl_generation"], "fer %.5f" % d.get("fer_end_to_end", -1), "teps %.1f" % d.get("mean_teps", 0), d.get("macro_batches"), (d.get("fer_vs_cpu") or {}).get("within_5_percent"))
PY
