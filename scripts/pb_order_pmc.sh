#!/bin/bash
# instruction-cache and wait counters of the PB chunk kernel under two frame orders.  usage: scripts/pb_order_pmc.sh <snr>  (GPU box)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; SNR=$1
cd /tmp
for ord in as longest; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_IFETCH --output-format csv -d $O/pbordpmc_${ord}_$SNR -- python3 $R/scripts/pb_order_experiment.py $SNR $ord > $O/pbordpmc_${ord}_$SNR.log 2>&1 || exit 1
  echo "== snr $SNR order $ord"
  python3 - $O/pbordpmc_${ord}_$SNR <<'PY'
import sys, glob, csv, collections, os
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(sys.argv[1], "**", "*_counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        k = row["Kernel_Name"]
        if "pb_wave" in k or "pb_coop" in k or "pb_singles" in k:
            acc[k.split("(")[0][:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, c in acc.items():
    print(k, {n: f"{sum(v[3:]) / max(1, len(v[3:])):.4g}" for n, v in c.items()})
PY
done
