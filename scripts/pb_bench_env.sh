#!/bin/bash
# graph-mode bench lines (the figure of merit) under environment settings.  usage: scripts/pb_bench_env.sh "snrs" "VAR=a" ...
SNRS=$1; shift
for E in "$@"; do for SNR in $SNRS; do
  env $E timeout -k 10 300 python bench.py --workload nms10_pb3 --snr $SNR --steps 20 --warmup 5 --no-cpu-baseline --no-overlap-pass > gpurun_out/tune.json 2>/dev/null || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/tune.json"))
print("$E snr $SNR frames/s %.4g ms/step %.3f pb %.4f" % (d["value"], d["ms_per_step"], [v for k,v in d["roofline"]["all_kernels_ms"].items() if k.startswith("pb_")][0]))
PY
done; done
