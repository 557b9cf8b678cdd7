#!/bin/bash
# tuning aid: PB-OSD bench lines for a grid of chunk targets.  usage: scripts/pb_tune.sh "T1 values" "T2 values" "snrs"
for T1 in $1; do for T2 in $2; do for SNR in $3; do
  LDPC_PB_T1=$T1 LDPC_PB_T2=$T2 timeout -k 10 200 python bench.py --workload nms10_pb3 --snr $SNR --steps 12 --warmup 4 --no-cpu-baseline --no-overlap-pass --no-graph > gpurun_out/tune.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/tune.json"))
print("t1 $T1 t2 $T2 snr $SNR ms/step %.3f pb %.4f" % (d["ms_per_step"], [v for k,v in d["roofline"]["all_kernels_ms"].items() if k.startswith("pb_")][0]))
PY
done; done; done
