#!/bin/bash
# round 4, pass 5: several finaliser workgroups -- tests, the fused launch by reduction mode again, fused vs petsc per config
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
python -m pytest tests/test_hip_pc.py tests/test_hip_parity.py -x -q 2>&1 | grep -v amdgpu | tail -6 | cut -c1-200 | tee gpurun_out/r4/pytest_5.log
for cfg in c3 c4 c5 c2; do
  python bench.py --micro-only --config $cfg --spmv-reps 100 2>&1 | grep "^micro" | tee -a gpurun_out/r4/micro_modes5.log
done
python bench.py --micro-only --rank-share 8 --spmv-reps 100 2>&1 | grep "^micro" | sed 's/^micro c3/micro c3s8/' | tee -a gpurun_out/r4/micro_modes5.log
python bench.py --micro-only --config c4 --rank-share 4 --spmv-reps 100 2>&1 | grep "^micro" | sed 's/^micro c4/micro c4s4/' | tee -a gpurun_out/r4/micro_modes5.log
python bench.py --micro-only --config c5 --rank-share 2 --spmv-reps 100 2>&1 | grep "^micro" | sed 's/^micro c5/micro c5s2/' | tee -a gpurun_out/r4/micro_modes5.log
run() { name=$1; shift; "$@" > gpurun_out/r4/$name.json 2> gpurun_out/r4/$name.log; }
for cfg in c3 c4 c5; do for m in fused petsc; do
  WAI_BCGS=$m run ${cfg}_${m}5 python bench.py --no-cpu --config $cfg
done; done
for m in fused petsc; do
  WAI_BCGS=$m run c3s8_${m}5 python bench.py --no-cpu --rank-share 8
done
python - <<'PY'
import json
for n in ["c3_fused5","c3_petsc5","c4_fused5","c4_petsc5","c5_fused5","c5_petsc5","c3s8_fused5","c3s8_petsc5"]:
    try:
        d=json.load(open("gpurun_out/r4/%s.json" % n)); c=d["config"]
        print("%-11s value %.3f acc %.2f its/step %.1f ms/it %.4f dev %s vec %s fused %.4f frac %.3f spmv %.3f fixed %.2f launches %.2f" % (n, d["value"], d.get("value_accepted_steps") or 0, c["krylov_iterations_per_newton_step"], c["ms_per_krylov_iteration"], c.get("ms_per_krylov_iteration_device_only"), c.get("ms_vector_updates_per_iteration"), d["roofline"]["ms_per_launch"], d["roofline"]["frac"], d["roofline"]["spmv_frac"], c["ms_fixed_per_newton_step"], c["launches_per_krylov_iteration"]))
    except Exception as e: print(n, e)
PY
python -m pytest tests/test_hip_multirank.py -x -q -k "shardings or two_ranks or tracer" 2>&1 | grep -v amdgpu | tail -6 | cut -c1-200 | tee gpurun_out/r4/pytest_multirank5.log
