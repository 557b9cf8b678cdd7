#!/bin/bash
# PB-OSD session on the GPU box: parity tests, then bench lines of nms10_pb3 at the given SNRs.  usage: scripts/gpu_pb_check.sh <tag> [snr ...]
TAG=${1:-pb}; shift
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_osd_pb.py tests/test_gpu_graph.py tests/test_gpu_streams.py -m gpu -x -q > $O/${TAG}_tests.log 2>&1
rc=$?; echo "[tests] rc=$rc"; tail -n 5 $O/${TAG}_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests hit their limit: stopping"; exit 1; fi
for SNR in "$@"; do
  timeout -k 10 300 python bench.py --workload nms10_pb3 --snr $SNR --steps 12 --warmup 4 --no-cpu-baseline --no-overlap-pass > $O/${TAG}_bench_snr$SNR.json 2> $O/${TAG}_bench_snr$SNR.err
  rc=$?; echo "[bench $SNR] rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "bench hit its limit: stopping"; exit 1; fi
  python - <<PY
import json
d=json.load(open("$O/${TAG}_bench_snr$SNR.json"))
print("snr", d["config"]["snr_db"], "frames/s %.4g" % d["value"], "ms/step %.3f" % d["ms_per_step"], "kernels", {k: round(v,4) for k,v in d["roofline"]["all_kernels_ms"].items()}, "mean_teps %.1f" % d["fer"]["mean_teps"], "fer_e2e %.5f" % d["fer"]["end_to_end_fer"])
PY
done
exit 0
