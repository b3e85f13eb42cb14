#!/bin/bash
# round-3 final GPU pass: the whole GPU suite, the evidence runs for c3 / c4 / c5, the rank shares on the same box
mkdir -p gpurun_out/r3
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then python -m pytest tests -m gpu -q -s --durations=10 > gpurun_out/r3/pytest_gpu_final.log 2>&1; fi
echo "pytest rc $?"; tail -14 gpurun_out/r3/pytest_gpu_final.log | cut -c1-160
for cfg in c3 c4 c5; do
  bash tools/gpu_profile.sh r3 $cfg --steps 20 --warmup 5 > gpurun_out/r3/gpu_profile_$cfg.log 2>&1
  head -6 gpurun_out/rocprof_kernel_stats_r3_$cfg.txt | cut -c1-140
done
share() { name=$1; shift
  python bench.py "$@" --steps 20 --warmup 5 --no-cpu > gpurun_out/bench_r3_$name.json 2> gpurun_out/bench_r3_$name.log
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o p -- python $R/bench.py "$@" --steps 20 --warmup 5 --no-cpu > /dev/null 2> $R/gpurun_out/r3/prof_$name.log)
  db=$(find /tmp/prof_$name -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/rocprof_summary.py "$db" gpurun_out/rocprof_kernel_stats_r3_$name.txt; fi
}
share c3_share8 --rank-share 8
share c4_share4 --config c4 --rank-share 4
share c5_share2 --config c5 --rank-share 2
python - <<'PY'
import json
for n in ["c3","c4","c5","c3_share8","c4_share4","c5_share2"]:
    try:
        d=json.load(open("gpurun_out/bench_r3_%s.json" % n)); c=d["config"]
        print("%-10s value %.3f acc %s its/step %.1f ms/it %.4f dev %s fused %.4f frac %.3f spmv %.3f fixed %.2f check %s" % (n, d["value"], d.get("value_accepted_steps"), c["krylov_iterations_per_newton_step"], c["ms_per_krylov_iteration"], c.get("ms_per_krylov_iteration_device_only"), d["roofline"]["ms_per_launch"], d["roofline"]["frac"], d["roofline"]["spmv_frac"], c["ms_fixed_per_newton_step"], d["check"].get("passed")))
    except Exception as e: print(n, e)
PY
