#!/bin/bash
# round-3 GPU pass 2: brick shapes at the rank share and at full size; c4 kernel stats
mkdir -p gpurun_out/r3
summ() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); c=d["config"]
print("%-40s value %.3f accepted %s its/step %.1f ms/it %.4f dev-only %s fused %.4f frac %.3f" % (sys.argv[1].split("/")[-1], d["value"], d.get("value_accepted_steps"), c["krylov_iterations_per_newton_step"], c["ms_per_krylov_iteration"], c.get("ms_per_krylov_iteration_device_only"), d["roofline"]["ms_per_launch"], d["roofline"]["frac"]))
PY
}
for b in "16 16 2" "12 12 3" "18 12 2" "12 12 2" "18 18 1"; do
  t=$(echo $b | tr ' ' x)
  python bench.py --rank-share 8 --brick $b --steps 20 --warmup 5 --no-cpu > gpurun_out/r3/scan_share8_$t.json 2> gpurun_out/r3/scan_share8_$t.log && summ gpurun_out/r3/scan_share8_$t.json
done
for b in "12 12 3" "18 12 2"; do
  t=$(echo $b | tr ' ' x)
  python bench.py --brick $b --steps 20 --warmup 5 --no-cpu > gpurun_out/r3/scan_c3_$t.json 2> gpurun_out/r3/scan_c3_$t.log && summ gpurun_out/r3/scan_c3_$t.json
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_c4 -o c4 -- python $GRAFT_REPO_ROOT/bench.py --config c4 --steps 10 --warmup 2 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/r3/bench_c4_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3/bench_c4_prof.log
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_c4 -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r3/c4_kernel_stats.csv; head -14 $f | cut -c1-200
summ gpurun_out/r3/bench_c4_prof.json
