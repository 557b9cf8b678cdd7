set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; B=$R/bench.py; O=$R/gpurun_out
cd /tmp
for w in nms10 nms10_osd0 nms10_fs2; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pw_$w -- python3 $B --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-overlap-pass > $O/pw_$w.log 2>&1
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pw_nms10_pb3 -- python3 $B --workload nms10_pb3 --steps 5 --warmup 1 --no-cpu-baseline --no-overlap-pass > $O/pw_nms10_pb3.log 2>&1
echo done
