#!/bin/bash
# round 4, pass 1: the three-launch BiCGStab iteration -- parity tests, then same-box A/B of the iteration forms at c3 and c3/8
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
python -m pytest tests/test_hip_pc.py -x -q -s 2>&1 | tail -25 | cut -c1-200 > gpurun_out/r4/pytest_pc.log
echo "pytest pc rc $?"; tail -5 gpurun_out/r4/pytest_pc.log
for m in fused petsc merged; do
  WAI_BCGS=$m python bench.py --no-cpu > gpurun_out/r4/c3_$m.json 2> gpurun_out/r4/c3_$m.log
done
for m in fused petsc; do
  WAI_BCGS=$m python bench.py --no-cpu --rank-share 8 > gpurun_out/r4/c3s8_$m.json 2> gpurun_out/r4/c3s8_$m.log
done
for cfg in c4 c5; do for m in fused petsc; do
  WAI_BCGS=$m python bench.py --no-cpu --config $cfg > gpurun_out/r4/${cfg}_$m.json 2> gpurun_out/r4/${cfg}_$m.log
done; done
python - <<'PY'
import json
for n in ["c3_fused","c3_petsc","c3_merged","c3s8_fused","c3s8_petsc","c4_fused","c4_petsc","c5_fused","c5_petsc"]:
    try:
        d=json.load(open("gpurun_out/r4/%s.json" % n)); c=d["config"]
        print("%-11s value %.3f acc %s its/step %.1f ms/it %.4f dev %s vec %s fused %.4f frac %.3f spmv %.3f fixed %.2f launches %.2f" % (n, d["value"], d.get("value_accepted_steps"), c["krylov_iterations_per_newton_step"], c["ms_per_krylov_iteration"], c.get("ms_per_krylov_iteration_device_only"), c.get("ms_vector_updates_per_iteration"), d["roofline"]["ms_per_launch"], d["roofline"]["frac"], d["roofline"]["spmv_frac"], c["ms_fixed_per_newton_step"], c["launches_per_krylov_iteration"]))
    except Exception as e: print(n, e)
PY
python -m pytest tests/test_hip_multirank.py -x -q 2>&1 | tail -25 | cut -c1-200 > gpurun_out/r4/pytest_multirank.log
echo "pytest multirank rc $?"; tail -8 gpurun_out/r4/pytest_multirank.log
