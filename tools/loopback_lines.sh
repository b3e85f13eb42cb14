# the BASELINE shardings at full size, all ranks on the one GPU of the box, over the stream-asynchronous test transport
# (tests/loopback_rccl/libasync_rccl.so): comm fields of the N > 1 bench line.   TAG=r5a tools/loopback_lines.sh "c5 2" "c4 4" "c3 8"
TAG=${TAG:-r5}
mkdir -p gpurun_out/$TAG
export WAI_RCCL_LIB=$PWD/tests/loopback_rccl/libasync_rccl.so WAI_BENCH_LOOPBACK=1
for spec in "$@"; do
  set -- $spec; cfg=$1; n=$2; shift 2     # the rest of the spec: extra bench arguments ("c3 8 --dt0 2000": no failed tries in the lead-in)
  MASTER_PORT=$((29500 + RANDOM % 500)) timeout ${LIMIT:-1200} python bench.py --config $cfg --gpus $n --lead 1 --steps ${STEPS:-3} --warmup 0 --no-cpu --spmv-reps 3 "$@" \
    > gpurun_out/$TAG/lb_${cfg}_$n.out 2> gpurun_out/$TAG/lb_${cfg}_$n.log
  echo "async loopback $cfg x $n rc $?"
  grep "^{" gpurun_out/$TAG/lb_${cfg}_$n.out > gpurun_out/bench_${TAG}_${cfg}_async$n.json
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/bench_${TAG}_${cfg}_async$n.json")); c=d["config"]
    print("$cfg x $n:", c["partition"], round(d["value"],4), c["krylov_iterations_per_newton_step"], d["comm"], d["check"])
except Exception as e: print("no line", e)
PY
done
