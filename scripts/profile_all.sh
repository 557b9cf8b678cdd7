#!/bin/bash
# profiles/rNN evidence for EVERY bench workload: kernel trace + the four PMC passes each.
#   scripts/profile_all.sh <tag>          (run on the GPU box; summaries: scripts/pmc_summary.py, see profiles/r03/README.md)
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; B=$R/bench.py; O=$R/gpurun_out; TAG=${1:-all}
cd /tmp
run() {   # name, bench args...
  local name=$1; shift
  local T=${TAG}_$name
  local A="--steps 10 --warmup 3 --no-cpu-baseline --no-overlap-pass --no-graph $*"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_stats -- python3 $B $A > $O/${T}_stats.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${T}_$c -- python3 $B $A > $O/${T}_$c.log 2>&1
  done
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/${T}_SQ -- python3 $B $A > $O/${T}_SQ.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/${T}_SQ2 -- python3 $B $A > $O/${T}_SQ2.log 2>&1
  echo "== $name"; python3 $R/scripts/kstats.py $O/${T}_stats
}
run nms10_osd2
run nms10 --workload nms10
run nms10_osd0 --workload nms10_osd0
run nms10_fs2 --workload nms10_fs2
run nms10_pb3 --workload nms10_pb3
run nms10_pb3_snr1.0 --workload nms10_pb3 --snr 1.0
echo "all profiles done"
