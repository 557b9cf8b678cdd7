#!/bin/bash
# serial (no graph) PB kernel times under environment settings.  usage: scripts/pb_kernel_times_env.sh <tag> <snr> "VAR=a" "VAR=b VAR2=c" ...
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=$1; SNR=$2; shift; shift
cd /tmp
i=0
for E in "$@"; do
  i=$((i+1))
  A="--workload nms10_pb3 --snr $SNR --steps 6 --warmup 2 --no-cpu-baseline --no-overlap-pass --no-graph"
  env $E timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_${i}_stats -- python3 $R/bench.py $A > $O/${TAG}_${i}_stats.log 2>&1 || exit 1
  echo "== $E snr $SNR"; python3 $R/scripts/kstats.py $O/${TAG}_${i}_stats | grep "pb_wave\|pb_coop\|pb_singles"
done
