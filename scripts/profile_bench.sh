#!/bin/bash
# profile capture for profiles/rNN (kernel trace + separate PMC passes); run on the GPU box:
#   scripts/profile_bench.sh <tag> [stats|all] [bench.py args...]
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; B=$R/bench.py; O=$R/gpurun_out; TAG=${1:-p}; MODE=${2:-all}; shift || true; shift || true
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- python3 $B --no-cpu-baseline --no-overlap-pass "$@" > $O/${TAG}_stats.log 2>&1
python3 $R/scripts/kstats.py $O/${TAG}_stats
[ "$MODE" = "stats" ] && exit 0
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${TAG}_$c -- python3 $B --steps 10 --warmup 2 --no-cpu-baseline --no-overlap-pass "$@" > $O/${TAG}_$c.log 2>&1
done
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/${TAG}_SQ -- python3 $B --steps 10 --warmup 2 --no-cpu-baseline --no-overlap-pass "$@" > $O/${TAG}_SQ.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/${TAG}_SQ2 -- python3 $B --steps 10 --warmup 2 --no-cpu-baseline --no-overlap-pass "$@" > $O/${TAG}_SQ2.log 2>&1
echo profiles done
