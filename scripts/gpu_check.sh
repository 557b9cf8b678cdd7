#!/bin/bash
# One GPU-box session: the -m gpu suite, then bench lines.  A step that is killed at its limit ends the session
# (no further GPU step after a hang).   usage: scripts/gpu_check.sh <tag> [pytest args...]
TAG=${1:-g}; shift
O=gpurun_out; mkdir -p $O
step() {   # name, limit, command...
  local name=$1 limit=$2; shift 2
  timeout -k 10 $limit "$@" > $O/${TAG}_$name.log 2>&1
  local rc=$?
  echo "[$name] rc=$rc"; tail -n 6 $O/${TAG}_$name.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[$name] hit its limit: stopping the session"; exit 1; fi
  return 0
}
step tests 1000 python -m pytest tests -m gpu -q -x "$@"
step bench 400 python bench.py
step bench_pb3 400 python bench.py --workload nms10_pb3 --steps 20 --warmup 3 --no-cpu-baseline
exit 0
