#!/bin/bash
# C4 with its new default bricks (8x4x2, k_pc_wave): full-size parity tests, then the evidence set, then its rank share
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_hip_fullsize.py -m gpu -x -q -s -k "c5" 2>&1 | tail -12 > gpurun_out/r3/run20_tests.log
cat gpurun_out/r3/run20_tests.log
bash tools/gpu_profile.sh r3 c5 --steps 20 --warmup 5 > gpurun_out/r3/run20_profile.log 2>&1
python bench.py --config c5 --rank-share 2 --no-cpu --steps 20 --warmup 5 > gpurun_out/bench_r3_c5_share2.json 2> gpurun_out/bench_r3_c5_share2.log
python - <<'PY'
import json
for n in ("c5", "c5_share2"):
    d = json.load(open("gpurun_out/bench_r3_%s.json" % n))
    r = d["roofline"]; c = d["config"]
    print(n, "value %.3f acc %s its/step %.1f ms/it %.4f fused %s frac %.3f spmv %.3f traffic %s check %s" % (d["value"], d.get("value_accepted_steps"), c["krylov_iterations_per_newton_step"], c["ms_per_krylov_iteration"], r.get("kernel"), r["frac"], r.get("spmv_frac", 0), r.get("traffic"), (d.get("check") or {}).get("passed")))
PY
