O=gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r03d_gpu_tests.log 2>&1; rc=$?; tail -n 3 $O/r03d_gpu_tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" || exit 1
b() { local name=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/r03d_$name.json 2> $O/r03d_$name.err; echo "[$name] rc=$?"; }
b driver --gpus 1 --steps 20 --warmup 5
b default
b pb3 --workload nms10_pb3
b pb3_snr1.0 --workload nms10_pb3 --snr 1.0 --steps 40 --warmup 4
python - <<PY
import json
for n in ("driver","default","pb3","pb3_snr1.0"):
    d=json.load(open("$O/r03d_%s.json"%n)); print(n, "%.4g"%d["value"], "%.4f"%d["ms_per_step"], {k:round(v,4) for k,v in d["roofline"]["all_kernels_ms"].items()}, d["roofline"]["traffic"], d["roofline"]["frac"])
PY
