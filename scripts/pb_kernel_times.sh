#!/bin/bash
# serial (no graph) PB-OSD kernel times from a rocprofv3 kernel trace.  usage: scripts/pb_kernel_times.sh <tag> <snr>   (GPU box)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=$1; SNR=$2
cd /tmp
A="--workload nms10_pb3 --snr $SNR --steps 6 --warmup 2 --no-cpu-baseline --no-overlap-pass --no-graph"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- python3 $R/bench.py $A > $O/${TAG}_stats.log 2>&1 || exit 1
echo "== snr $SNR"; python3 $R/scripts/kstats.py $O/${TAG}_stats | grep "pb_wave\|pb_coop\|pb_singles"
