#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -k "gmres or ksp or krylov or solver or pc" -p no:cacheprovider -x 2>&1 | tail -5 > gpurun_out/gmres_tests_r6.log
cat gpurun_out/gmres_tests_r6.log
python -m pytest tests/test_hip_parity.py -q -m gpu -p no:cacheprovider 2>&1 | tail -3
for cfg in c3 c2; do
python bench.py --config $cfg --ksp gmres --no-cpu --steps 10 --warmup 3 > gpurun_out/bench_r6b_${cfg}_gmres.json 2> gpurun_out/bench_r6b_${cfg}_gmres.log
python - <<P
import json
g=json.load(open("gpurun_out/bench_r6b_${cfg}_gmres.json")); r=g["roofline"]
print("$cfg gmres: value %.3f, %.4f ms/iteration, %.1f its/step; by kernel %s; dominant %s frac %.3f" % (g["value"], g["config"]["ms_per_krylov_iteration"], g["config"]["krylov_iterations_per_newton_step"], r["ms_per_krylov_iteration_by_kernel"], r["dominant_half"], r["frac"]))
P
done
