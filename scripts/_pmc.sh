#!/bin/bash
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; B=$R/bench.py; O=$R/gpurun_out; TAG=${1:-q}
cd /tmp
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/${TAG}_SQ -- python3 $B --steps 10 --warmup 2 --no-cpu-baseline --no-overlap-pass > $O/${TAG}_SQ.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/${TAG}_SQ2 -- python3 $B --steps 10 --warmup 2 --no-cpu-baseline --no-overlap-pass > $O/${TAG}_SQ2.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $O/${TAG}_SQ3 -- python3 $B --steps 10 --warmup 2 --no-cpu-baseline --no-overlap-pass > $O/${TAG}_SQ3.log 2>&1 || true
echo pmc done
