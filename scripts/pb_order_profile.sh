#!/bin/bash
# per-kernel times of the PB-OSD launch under three frame orders (rocprofv3 kernel trace).  usage: scripts/pb_order_profile.sh <snr>  (GPU box)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; SNR=$1
cd /tmp
for ord in as longest shortest; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pbord_${ord}_$SNR -- python3 $R/scripts/pb_order_experiment.py $SNR $ord > $O/pbord_${ord}_$SNR.log 2>&1 || exit 1
  echo "== snr $SNR order $ord"; python3 $R/scripts/kstats.py $O/pbord_${ord}_$SNR | grep "pb_wave\|pb_coop\|pb_singles\|pb_seq"
done
