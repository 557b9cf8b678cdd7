#!/bin/bash
# where do PB-OSD's long searches stand: TEP histogram, serial (no graph) kernel times.  usage: scripts/pb_longsearch_probe.sh <tag> [snr ...]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=$1; shift
mkdir -p $O
timeout -k 10 200 python3 $R/scripts/pb_ntep_hist.py "$@" > $O/${TAG}_hist.log 2>&1 || exit 1
cat $O/${TAG}_hist.log
cd /tmp
for SNR in "$@"; do
  A="--workload nms10_pb3 --snr $SNR --steps 6 --warmup 2 --no-cpu-baseline --no-overlap-pass --no-graph"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_snr${SNR}_stats -- python3 $R/bench.py $A > $O/${TAG}_snr${SNR}_stats.log 2>&1 || exit 1
  echo "== $SNR"; python3 $R/scripts/kstats.py $O/${TAG}_snr${SNR}_stats | grep "pb_"
done
