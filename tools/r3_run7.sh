#!/bin/bash
# round-3 GPU pass 7: the batched finaliser -- parity subset, rank shares, c3 with kernel stats
mkdir -p gpurun_out/r3
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_hip_parity.py tests/test_hip_pc.py -m gpu -x -q 2>&1 | tail -2
line() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); c=d["config"]
print(sys.argv[1].split("/")[-1], "value %.3f its/step %.1f ms/it %.4f dev-only %s fused %.4f vectors %s" % (d["value"], c["krylov_iterations_per_newton_step"], c["ms_per_krylov_iteration"], c.get("ms_per_krylov_iteration_device_only"), d["roofline"]["ms_per_launch"], c.get("ms_vector_updates_per_iteration")))
PY
}
python bench.py --rank-share 8 --steps 20 --warmup 5 --no-cpu > gpurun_out/r3/b7_share8.json 2> gpurun_out/r3/b7_share8.log; line gpurun_out/r3/b7_share8.json
python bench.py --config c4 --rank-share 4 --steps 20 --warmup 5 --no-cpu > gpurun_out/r3/b7_c4share4.json 2> gpurun_out/r3/b7_c4share4.log; line gpurun_out/r3/b7_c4share4.json
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_b7 -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu > $R/gpurun_out/r3/b7_c3.json 2> $R/gpurun_out/r3/b7_c3.log)
db=$(find /tmp/prof_b7 -name "*.db" | head -1)
if [ -n "$db" ]; then python tools/rocprof_summary.py "$db" gpurun_out/r3/kstats_b7_c3.txt; head -8 gpurun_out/r3/kstats_b7_c3.txt | cut -c1-140; fi
line gpurun_out/r3/b7_c3.json
