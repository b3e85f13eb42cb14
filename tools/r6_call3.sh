#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -k "gmres or ksp or krylov or solver" -p no:cacheprovider -x 2>&1 | tail -3
for cfg in c3 c2; do
python bench.py --config $cfg --ksp gmres --no-cpu --steps 10 --warmup 3 > gpurun_out/bench_r6c_${cfg}_gmres.json 2> gpurun_out/bench_r6c_${cfg}_gmres.log
python - <<P
import json
g=json.load(open("gpurun_out/bench_r6c_${cfg}_gmres.json")); r=g["roofline"]
print("$cfg gmres: value %.3f, %.4f ms/iteration, %.1f its/step; by kernel %s; dominant %s frac %.3f" % (g["value"], g["config"]["ms_per_krylov_iteration"], g["config"]["krylov_iterations_per_newton_step"], r["ms_per_krylov_iteration_by_kernel"], r["dominant_half"], r["frac"]))
P
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L > gpurun_out/rocprofv3_counters_list.txt 2>&1
grep -oE "\b(TCC|TCP|TA|TD|TCA)_[A-Z_0-9a-z\[\]]+" gpurun_out/rocprofv3_counters_list.txt | sort -u | tr '\n' ' ' | cut -c1-6000
