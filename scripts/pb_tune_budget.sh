#!/bin/bash
# tuning aid: PB-OSD bench lines for a grid of hand-off budgets.  usage: scripts/pb_tune_budget.sh "S:M:L:MAXLEN ..." "snrs"
for B in $1; do IFS=: read S M L ML <<< "$B"; for SNR in $2; do
  LDPC_PB_BUDGET_S=$S LDPC_PB_BUDGET_M=$M LDPC_PB_BUDGET=$L LDPC_PB_HANDOFF_MAXLEN=$ML timeout -k 10 200 python bench.py --workload nms10_pb3 --snr $SNR --steps 12 --warmup 4 --no-cpu-baseline --no-overlap-pass --no-graph > gpurun_out/tune.json 2>/dev/null || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/tune.json"))
print("budgets $B snr $SNR ms/step %.3f pb %.4f" % (d["ms_per_step"], [v for k,v in d["roofline"]["all_kernels_ms"].items() if k.startswith("pb_")][0]))
PY
done; done
