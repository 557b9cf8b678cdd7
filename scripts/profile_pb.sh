#!/bin/bash
# PB-OSD evidence for profiles/rNN: kernel trace + PMC passes of `bench.py --workload nms10_pb3` at the given SNRs
#   scripts/profile_pb.sh <tag> <snr> [<snr> ...]
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; B=$R/bench.py; O=$R/gpurun_out; TAG=${1:-pb}; shift
cd /tmp
for SNR in "$@"; do
  T=${TAG}_snr${SNR}
  A="--workload nms10_pb3 --snr $SNR --steps 6 --warmup 2 --no-cpu-baseline --no-overlap-pass --no-graph"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_stats -- python3 $B $A > $O/${T}_stats.log 2>&1
  python3 $R/scripts/kstats.py $O/${T}_stats
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/${T}_SQ -- python3 $B $A > $O/${T}_SQ.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/${T}_SQ2 -- python3 $B $A > $O/${T}_SQ2.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${T}_$c -- python3 $B $A > $O/${T}_$c.log 2>&1
  done
  echo "profiles of $T done"
done
