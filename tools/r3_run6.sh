#!/bin/bash
# round-3 GPU pass 6: the rank-share lines (one rank's share of the N-GPU decomposition on one GPU) with kernel stats
mkdir -p gpurun_out/r3
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
share() { name=$1; shift
  python bench.py "$@" --steps 20 --warmup 5 --no-cpu > gpurun_out/bench_r3_$name.json 2> gpurun_out/bench_r3_$name.log
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o p -- python $R/bench.py "$@" --steps 20 --warmup 5 --no-cpu > /dev/null 2> $R/gpurun_out/r3/prof_$name.log)
  db=$(find /tmp/prof_$name -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/rocprof_summary.py "$db" gpurun_out/rocprof_kernel_stats_r3_$name.txt; head -8 gpurun_out/rocprof_kernel_stats_r3_$name.txt | cut -c1-140; fi
  python - "$name" <<'PY'
import json,sys
d=json.load(open("gpurun_out/bench_r3_%s.json" % sys.argv[1])); c=d["config"]
print(sys.argv[1], "value %.3f its/step %.1f ms/it %.4f dev-only %.4f fused %.4f vectors %.4f launches %.2f" % (d["value"], c["krylov_iterations_per_newton_step"], c["ms_per_krylov_iteration"], c["ms_per_krylov_iteration_device_only"], d["roofline"]["ms_per_launch"], c["ms_vector_updates_per_iteration"], c["launches_per_krylov_iteration"]))
PY
}
share c3_share8 --rank-share 8
share c4_share4 --config c4 --rank-share 4
share c5_share2 --config c5 --rank-share 2
