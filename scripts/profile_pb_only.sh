#!/bin/bash
# scripts/profile_all.sh for the two PB-OSD workloads only (kernel trace + the four PMC passes each), then their bench lines.
#   scripts/profile_pb_only.sh <tag>          (GPU box)
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; B=$R/bench.py; O=$R/gpurun_out; TAG=${1:-pbonly}
cd /tmp
run() {
  local name=$1; shift
  local T=${TAG}_$name
  local A="--steps 10 --warmup 3 --no-cpu-baseline --no-overlap-pass --no-graph $*"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_stats -- python3 $B $A > $O/${T}_stats.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${T}_$c -- python3 $B $A > $O/${T}_$c.log 2>&1
  done
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/${T}_SQ -- python3 $B $A > $O/${T}_SQ.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/${T}_SQ2 -- python3 $B $A > $O/${T}_SQ2.log 2>&1
  echo "== $name"; python3 $R/scripts/kstats.py $O/${T}_stats | grep "pb_"
}
run nms10_pb3 --workload nms10_pb3
run nms10_pb3_snr1.0 --workload nms10_pb3 --snr 1.0
cd $R
for w in nms10_pb3 nms10_pb3_snr1.0; do python3 scripts/pmc_summary.py $O/${TAG}_$w profiles/r03/pmc_counters_${w}_r03.json --workload $w > /dev/null; cp profiles/r03/pmc_counters_${w}_r03.json $O/${TAG}_pmc_${w}.json; done
timeout -k 10 400 python bench.py --workload nms10_pb3 > $O/${TAG}_pb3.json 2> $O/${TAG}_pb3.err; echo "[pb3] rc=$?"
timeout -k 10 400 python bench.py --workload nms10_pb3 --snr 1.0 --steps 40 --warmup 4 > $O/${TAG}_pb3_snr1.0.json 2> $O/${TAG}_pb3_snr1.0.err; echo "[pb3 1.0] rc=$?"
python3 - <<PY
import json
for n in ("pb3","pb3_snr1.0"):
    d=json.load(open("$O/${TAG}_%s.json"%n)); print(n, "%.4g"%d["value"], "%.4f"%d["ms_per_step"], {k:round(v,4) for k,v in d["roofline"]["all_kernels_ms"].items()}, d["roofline"]["traffic"])
PY
