#!/bin/bash
# bench.py --workload nms10_pb3 (graph, four batches in flight) under PB-OSD tunings.  usage: scripts/bench_pb_tunings.sh <snr> <tuning> [<tuning> ...]  ("-" = defaults)   (GPU box)
SNR=$1; shift
for t in "$@"; do
  if [ "$t" = "-" ]; then A=""; else A="--pb-tuning $t"; fi
  for rep in 1 2; do
    timeout -k 10 200 python bench.py --workload nms10_pb3 --snr $SNR --steps 40 --warmup 8 --no-cpu-baseline --no-overlap-pass $A 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('snr $SNR tuning $t', '%.4g frames/s' % d['value'], '%.4f ms/step' % d['ms_per_step'])" || exit 1
  done
done
