#!/bin/bash
mkdir -p gpurun_out/r3
python -m pytest tests -m gpu -q --durations=6 > gpurun_out/r3/pytest_gpu_last.log 2>&1; echo "pytest rc $?"; tail -9 gpurun_out/r3/pytest_gpu_last.log | cut -c1-160
python bench.py --config c2 --steps 20 --warmup 5 > gpurun_out/bench_r3_c2.json 2> gpurun_out/bench_r3_c2.log
python bench.py --config c2 --steps 20 --warmup 5 --ksp gmres --no-cpu > gpurun_out/bench_r3_c2_gmres.json 2> gpurun_out/bench_r3_c2_gmres.log
python - <<'PY'
import json
for n in ["c2","c2_gmres"]:
    try:
        d=json.load(open("gpurun_out/bench_r3_%s.json" % n)); c=d["config"]
        print("%-10s value %.3f acc %s its/step %.1f ms/it %.4f fused %.4f frac %.3f spmv %.3f check %s cpu %s" % (n, d["value"], d.get("value_accepted_steps"), c["krylov_iterations_per_newton_step"], c["ms_per_krylov_iteration"], d["roofline"]["ms_per_launch"], d["roofline"]["frac"], d["roofline"]["spmv_frac"], d["check"].get("passed"), d.get("cpu_baseline",{}).get("value")))
    except Exception as e: print(n, e)
PY
