#!/bin/bash
# round 4, pass 2: stored-S fused iteration (4 launches) and the wave-staged sweeps of k_pc_park -- tests, then same-box A/B
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
python -m pytest tests/test_hip_pc.py tests/test_hip_parity.py -x -q 2>&1 | tail -15 | cut -c1-200 > gpurun_out/r4/pytest_pc2.log
echo "pytest pc+parity rc $?"; tail -3 gpurun_out/r4/pytest_pc2.log
run() { name=$1; shift; "$@" > gpurun_out/r4/$name.json 2> gpurun_out/r4/$name.log; }
for ws in 1 0; do
  WAI_PC_WAVESTAGE=$ws run c3_ws$ws python bench.py --no-cpu
  WAI_PC_WAVESTAGE=$ws run c3s8_ws$ws python bench.py --no-cpu --rank-share 8
  WAI_PC_WAVESTAGE=$ws run c2_ws$ws python bench.py --no-cpu --config c2
done
WAI_BCGS=petsc run c3_petsc2 python bench.py --no-cpu
WAI_BCGS=petsc run c3s8_petsc2 python bench.py --no-cpu --rank-share 8
run c4_fused2 python bench.py --no-cpu --config c4
run c5_fused2 python bench.py --no-cpu --config c5
python - <<'PY'
import json
for n in ["c3_ws1","c3_ws0","c3_petsc2","c3s8_ws1","c3s8_ws0","c3s8_petsc2","c2_ws1","c2_ws0","c4_fused2","c5_fused2"]:
    try:
        d=json.load(open("gpurun_out/r4/%s.json" % n)); c=d["config"]
        print("%-11s value %.3f acc %.2f its/step %.1f ms/it %.4f dev %s vec %s fused %.4f frac %.3f spmv %.3f fixed %.2f launches %.2f" % (n, d["value"], d.get("value_accepted_steps") or 0, c["krylov_iterations_per_newton_step"], c["ms_per_krylov_iteration"], c.get("ms_per_krylov_iteration_device_only"), c.get("ms_vector_updates_per_iteration"), d["roofline"]["ms_per_launch"], d["roofline"]["frac"], d["roofline"]["spmv_frac"], c["ms_fixed_per_newton_step"], c["launches_per_krylov_iteration"]))
    except Exception as e: print(n, e)
PY
python -m pytest tests/test_hip_multirank.py -x -q 2>&1 | grep -v amdgpu.ids | tail -25 | cut -c1-200 > gpurun_out/r4/pytest_multirank2.log
echo "pytest multirank rc $?"; tail -8 gpurun_out/r4/pytest_multirank2.log
