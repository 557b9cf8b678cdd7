#!/bin/bash
# quick PMC look at the PB kernels: one stats pass + the two SQ passes.  usage: scripts/profile_pb_quick.sh <tag> <snr>
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; B=$R/bench.py; O=$R/gpurun_out; TAG=$1; SNR=$2
cd /tmp
T=${TAG}_snr${SNR}
A="--workload nms10_pb3 --snr $SNR --steps 6 --warmup 2 --no-cpu-baseline --no-overlap-pass --no-graph"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_stats -- python3 $B $A > $O/${T}_stats.log 2>&1
python3 $R/scripts/kstats.py $O/${T}_stats | grep "pb_"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/${T}_SQ -- python3 $B $A > $O/${T}_SQ.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/${T}_SQ2 -- python3 $B $A > $O/${T}_SQ2.log 2>&1
python3 $R/scripts/pmc_summary.py $O/${T} $O/${T}_pmc.json --workload nms10_pb3_snr$SNR > /dev/null
python3 - <<PY
import json
d=json.load(open("$O/${T}_pmc.json"))
for k,v in d["per_launch_mean"].items():
    if "pb_wave" in k or "pb_coop" in k:
        wc=v["SQ_WAVE_CYCLES"]; print(k); print({c:round(x/1e6,1) for c,x in v.items() if c.startswith("SQ")})
        print("wait_any %.2f wait_inst %.2f valu_active/wave_cycles %.2f  valu_issue_frac(per SIMD) %.2f" % (v["SQ_WAIT_ANY"]/wc, v["SQ_WAIT_INST_ANY"]/wc, v["SQ_ACTIVE_INST_VALU"]/wc, v.get("valu_issue_frac",0)))
PY
