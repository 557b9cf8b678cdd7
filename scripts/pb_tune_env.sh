#!/bin/bash
# tuning aid: PB-OSD bench lines under environment settings.  usage: scripts/pb_tune_env.sh "snrs" "VAR=a VAR2=b" "VAR=c" ...
SNRS=$1; shift
for E in "$@"; do for SNR in $SNRS; do
  env $E timeout -k 10 200 python bench.py --workload nms10_pb3 --snr $SNR --steps 12 --warmup 4 --no-cpu-baseline --no-overlap-pass --no-graph > gpurun_out/tune.json 2>/dev/null || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/tune.json"))
print("$E snr $SNR ms/step %.3f pb %.4f" % (d["ms_per_step"], [v for k,v in d["roofline"]["all_kernels_ms"].items() if k.startswith("pb_")][0]))
PY
done; done
