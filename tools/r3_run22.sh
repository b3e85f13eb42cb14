#!/bin/bash
# the BASELINE shardings at full size over the one-GPU loopback transport (tests/loopback_rccl): the multi-rank code path
# end to end at size; timings mean nothing (N processes time-slice one GPU and exchange through host memory).
# usage: bash tools/r3_run22.sh <config> <ranks> [<config> <ranks> ...]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3
export WAI_RCCL_LIB=$PWD/tests/loopback_rccl/libloopback_rccl.so WAI_BENCH_LOOPBACK=1
while [ $# -ge 2 ]; do
  cfg=$1; n=$2; shift 2
  MASTER_PORT=$((29500 + RANDOM % 500)) timeout 1500 python bench.py --config $cfg --gpus $n --steps ${STEPS:-8} --warmup ${WARM:-2} --no-cpu --spmv-reps 5 > gpurun_out/bench_r3_${cfg}_loopback$n.out 2> gpurun_out/bench_r3_${cfg}_loopback$n.log
  echo "rc $?"
  grep "^{" gpurun_out/bench_r3_${cfg}_loopback$n.out > gpurun_out/bench_r3_${cfg}_loopback$n.json
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/bench_r3_${cfg}_loopback$n.json")); c=d["config"]
    print("$cfg x $n:", d["n_gpus"], c["partition"], round(d["value"],4), c["krylov_iterations_per_newton_step"], round(c["ms_per_krylov_iteration"],2), [(x["time_step"], x["newton"], x["krylov"], x["reason"]) for x in c["timed_newton_steps"]])
except Exception as e: print("no line", e)
PY
done
