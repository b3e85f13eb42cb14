#!/bin/bash
# round-4 final GPU pass: the whole GPU suite, the evidence runs for c3 / c4 / c5 (bench line, rocprofv3 kernel stats, PMC
# traffic), the rank shares on the same box, the BASELINE shardings at full size over the loopback (comm fields)
mkdir -p gpurun_out/r4
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r4}   # names of the evidence files: bench_${TAG}_c3.json ... (r4c: the pass on the round's last commit)
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then python -m pytest tests -m gpu -q -s --durations=10 > gpurun_out/r4/pytest_gpu_final.log 2>&1; fi
echo "pytest rc $?"; grep -v amdgpu gpurun_out/r4/pytest_gpu_final.log | tail -14 | cut -c1-160
for cfg in ${CFGS:-c3 c4 c5}; do
  bash tools/gpu_profile.sh $TAG $cfg --steps 20 --warmup 5 > gpurun_out/r4/gpu_profile_$cfg.log 2>&1
  head -6 gpurun_out/rocprof_kernel_stats_${TAG}_$cfg.txt | cut -c1-140
done
share() { name=$1; shift
  python bench.py "$@" --steps 20 --warmup 5 --no-cpu > gpurun_out/bench_${TAG}_$name.json 2> gpurun_out/bench_${TAG}_$name.log
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o p -- python $R/bench.py "$@" --steps 20 --warmup 5 --no-cpu > /dev/null 2> $R/gpurun_out/r4/prof_$name.log)
  db=$(find /tmp/prof_$name -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/rocprof_summary.py "$db" gpurun_out/rocprof_kernel_stats_${TAG}_$name.txt > /dev/null; fi
}
if [ -z "$SKIP_SHARES" ]; then
share c3_share8 --rank-share 8
share c4_share4 --config c4 --rank-share 4
share c5_share2 --config c5 --rank-share 2
python bench.py --config c2 > gpurun_out/bench_${TAG}_c2.json 2> gpurun_out/bench_${TAG}_c2.log
fi
TAG=$TAG python - <<'PY'
import json, os
TAG = os.environ["TAG"]
for n in ["c3","c4","c5","c2","c3_share8","c4_share4","c5_share2"]:
    try:
        d=json.load(open("gpurun_out/bench_%s_%s.json" % (TAG, n))); c=d["config"]
        print("%-10s value %.3f acc %s its/step %.1f ms/it %.4f dev %s fused %.4f frac %.3f spmv %.3f fixed %.2f check %s cpu %s" % (n, d["value"], d.get("value_accepted_steps"), c["krylov_iterations_per_newton_step"], c["ms_per_krylov_iteration"], c.get("ms_per_krylov_iteration_device_only"), d["roofline"]["ms_per_launch"], d["roofline"]["frac"], d["roofline"]["spmv_frac"], c["ms_fixed_per_newton_step"], d["check"].get("passed"), (d.get("cpu_baseline") or {}).get("value")))
    except Exception as e: print(n, e)
PY
if [ -z "$SKIP_LOOPBACK" ]; then
export WAI_RCCL_LIB=$PWD/tests/loopback_rccl/libloopback_rccl.so WAI_BENCH_LOOPBACK=1 WAI_HALO_OVERLAP=0
lb() { cfg=$1; n=$2
  MASTER_PORT=$((29500 + RANDOM % 500)) timeout 1200 python bench.py --config $cfg --gpus $n --lead 1 --steps 3 --warmup 0 --no-cpu --spmv-reps 3 > gpurun_out/r4/lb_${cfg}_$n.out 2> gpurun_out/r4/lb_${cfg}_$n.log
  echo "loopback $cfg x $n rc $?"
  grep "^{" gpurun_out/r4/lb_${cfg}_$n.out > gpurun_out/bench_${TAG}_${cfg}_loopback$n.json
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/bench_${TAG}_${cfg}_loopback$n.json")); c=d["config"]
    print("$cfg x $n:", c["partition"], round(d["value"],4), c["krylov_iterations_per_newton_step"], d["comm"], d["check"])
except Exception as e: print("no line", e)
PY
}
lb c5 2
lb c4 4
lb c3 8
fi
