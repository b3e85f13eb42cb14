#!/bin/bash
# evidence pass of a round (ONE gpurun call = one box; TAG names the files: bench_${TAG}_c3.json ...): the whole GPU suite, the evidence runs for c3 / c4 / c5 (bench line, rocprofv3 kernel stats, PMC
# traffic), the rank shares on the same box, the BASELINE shardings at full size over the loopback (comm fields)
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r6}
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then python -m pytest tests -m gpu -q --durations=10 -p no:cacheprovider > gpurun_out/$TAG/pytest_gpu_final.log 2>&1; fi
echo "pytest rc $?"; grep -v amdgpu gpurun_out/$TAG/pytest_gpu_final.log | tail -14 | cut -c1-160
for cfg in ${CFGS:-c3 c4 c5}; do
  bash tools/gpu_profile.sh $TAG $cfg --steps 20 --warmup 5 > gpurun_out/$TAG/gpu_profile_$cfg.log 2>&1
  head -6 gpurun_out/rocprof_kernel_stats_${TAG}_$cfg.txt | cut -c1-140
done
share() { name=$1; shift
  python bench.py "$@" --steps 20 --warmup 5 --no-cpu > gpurun_out/bench_${TAG}_$name.json 2> gpurun_out/bench_${TAG}_$name.log
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o p -- python $R/bench.py "$@" --steps 20 --warmup 5 --no-cpu > /dev/null 2> $R/gpurun_out/$TAG/prof_$name.log)
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
# BASELINE's shardings at full size, all ranks on this one GPU over the stream-asynchronous test transport
TAG=$TAG bash tools/loopback_lines.sh "c5 2" "c4 4" "c3 2"
fi
if [ -z "$SKIP_SERIES" ]; then
# SURVEY.md section 8d's second series: Corey (0.3, 0.05) curves, and GMRES(30) beside BiCGStab, at C2 and C3
for cfg in c2 c3; do
  python bench.py --config $cfg --curves corey --no-cpu > gpurun_out/bench_${TAG}_${cfg}_corey.json 2> gpurun_out/$TAG/bench_${cfg}_corey.log
  python bench.py --config $cfg --ksp gmres --no-cpu --steps 10 --warmup 3 > gpurun_out/bench_${TAG}_${cfg}_gmres.json 2> gpurun_out/$TAG/bench_${cfg}_gmres.log
done
TAG=$TAG python - <<'PY'
import json, os
TAG = os.environ["TAG"]
for n in ["c2_corey","c3_corey","c2_gmres","c3_gmres"]:
    try:
        d=json.load(open("gpurun_out/bench_%s_%s.json" % (TAG, n))); c=d["config"]
        print("%-10s value %.3f norm %s its/step %.1f ms/it %.4f fixed %.2f" % (n, d["value"], d.get("value_normalised"), c["krylov_iterations_per_newton_step"], c["ms_per_krylov_iteration"], c["ms_fixed_per_newton_step"]))
    except Exception as e: print(n, e)
PY
fi
