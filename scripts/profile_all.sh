#!/bin/bash
# profiles/rNN evidence for EVERY bench workload: kernel trace (--stats) + the four PMC passes each (separate runs, --kernel-trace
# only beside --pmc), folded into gpurun_out/<round>/kernel_stats_<workload>_<round>.csv and pmc_counters_<workload>_<round>.json
# (copied into profiles/<round>/ once they are back from the GPU box).
#   scripts/profile_all.sh <round, e.g. r04> [workload names ...]          (run on the GPU box; default: all)
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; B=$R/bench.py; O=$R/gpurun_out; RND=${1:-r04}; shift || true
P=$O/$RND; mkdir -p $P $O     # (only gpurun_out/ travels back from the GPU box: copy gpurun_out/<round>/* into profiles/<round>/ afterwards)
cd /tmp
run() {   # name, frames per launch, bench args...
  local name=$1 frames=$2; shift 2
  local T=${RND}_$name
  local A="--steps 10 --warmup 3 --no-cpu-baseline --no-overlap-pass --no-graph $*"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_stats -- python3 $B $A > $O/${T}_stats.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${T}_$c -- python3 $B $A > $O/${T}_$c.log 2>&1
  done
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/${T}_SQ -- python3 $B $A > $O/${T}_SQ.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/${T}_SQ2 -- python3 $B $A > $O/${T}_SQ2.log 2>&1
  echo "== $name"; python3 $R/scripts/kstats.py $O/${T}_stats
  cp $(ls $O/${T}_stats/*/*kernel_stats.csv | head -1) $P/kernel_stats_${name}_${RND}.csv
  python3 $R/scripts/pmc_summary.py $O/$T $P/pmc_counters_${name}_${RND}.json --workload $name --frames $frames > /dev/null
}
want() { [ $# -eq 0 ] && return 0; local n=$1; shift; for w in "$@"; do [ "$w" = "$n" ] && return 0; done; return 1; }
SEL="$@"
sel() { if [ -z "$SEL" ]; then return 0; fi; for w in $SEL; do [ "$w" = "$1" ] && return 0; done; return 1; }
sel nms10_osd2 && run nms10_osd2 131072
sel nms10_osd2_fused && run nms10_osd2_fused 131072 --osd-route decode
sel nms10 && run nms10 65536 --workload nms10
sel nms10_osd0 && run nms10_osd0 65536 --workload nms10_osd0
sel nms10_fs2 && run nms10_fs2 131072 --workload nms10_fs2
sel nms10_pb3 && run nms10_pb3 131072 --workload nms10_pb3
sel nms10_pb3_snr1.0 && run nms10_pb3_snr1.0 131072 --workload nms10_pb3 --snr 1.0
sel surface_nms && run surface_nms 131072 --workload surface_nms --steps 4 --warmup 2
echo "all profiles done"
