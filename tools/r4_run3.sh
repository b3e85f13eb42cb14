#!/bin/bash
# round 4, pass 3: finaliser that sums the five merged reductions together (fused vs petsc, same box, all configs); the
# LDS-tiled residual and the store-only Jacobian -- identity tests, durations and traffic at c3 and c5
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
python -m pytest tests/test_hip_pc.py tests/test_hip_parity.py tests/test_hip_fullsize.py -x -q 2>&1 | grep -v amdgpu | tail -15 | cut -c1-200 > gpurun_out/r4/pytest_3.log
echo "pytest rc $?"; tail -3 gpurun_out/r4/pytest_3.log
run() { name=$1; shift; "$@" > gpurun_out/r4/$name.json 2> gpurun_out/r4/$name.log; }
for cfg in c3 c4 c5; do for m in fused petsc; do
  WAI_BCGS=$m run ${cfg}_${m}3 python bench.py --no-cpu --config $cfg
done; done
for m in fused petsc; do
  WAI_BCGS=$m run c3s8_${m}3 python bench.py --no-cpu --rank-share 8
done
python - <<'PY'
import json
for n in ["c3_fused3","c3_petsc3","c4_fused3","c4_petsc3","c5_fused3","c5_petsc3","c3s8_fused3","c3s8_petsc3"]:
    try:
        d=json.load(open("gpurun_out/r4/%s.json" % n)); c=d["config"]
        print("%-11s value %.3f acc %.2f its/step %.1f ms/it %.4f dev %s vec %s fused %.4f frac %.3f spmv %.3f fixed %.2f launches %.2f" % (n, d["value"], d.get("value_accepted_steps") or 0, c["krylov_iterations_per_newton_step"], c["ms_per_krylov_iteration"], c.get("ms_per_krylov_iteration_device_only"), c.get("ms_vector_updates_per_iteration"), d["roofline"]["ms_per_launch"], d["roofline"]["frac"], d["roofline"]["spmv_frac"], c["ms_fixed_per_newton_step"], c["launches_per_krylov_iteration"]))
    except Exception as e: print(n, e)
PY
bash tools/asm_traffic.sh r4 c3
bash tools/asm_traffic.sh r4 c5
python -m pytest tests/test_hip_multirank.py -x -q -k "asm_overlap" 2>&1 | grep -v amdgpu | tail -8 | cut -c1-200 > gpurun_out/r4/pytest_multirank3.log
tail -3 gpurun_out/r4/pytest_multirank3.log
