#!/bin/bash
# round 6: the reference's DEFAULT preconditioner (PCASM overlap 1 / ILU(0), src/timestepper.F90:2019-2020) against the
# brick block-Jacobi this library defaults to, on the SAME linear system (the window's first: after the lead-in), and at the
# full protocol at C2: Krylov count and time per solve.
mkdir -p gpurun_out
L=gpurun_out/pc_compare_r6.log; : > $L
for cfg in c2 c3; do
  for pc in bjacobi asm; do
    echo "== $cfg --pc $pc (first system of the window)" >> $L
    WAI_MICRO_LEAD=1 timeout 900 python bench.py --config $cfg --pc $pc --micro-only 2>&1 | grep -E "first system|Error|error" | cut -c1-400 >> $L
  done
done
for pc in bjacobi asm; do
  echo "== c2 --pc $pc (full protocol, 20 timed Newton steps)" >> $L
  timeout 1200 python bench.py --config c2 --pc $pc --no-cpu > gpurun_out/bench_r6_c2_pc_$pc.json 2> gpurun_out/bench_r6_c2_pc_$pc.log
  python - <<P >> $L
import json
d = json.load(open("gpurun_out/bench_r6_c2_pc_$pc.json")); c = d["config"]
print("value %.3f Newton steps/s, normalised %s, %.1f Krylov its / Newton step, %.4f ms / iteration, fixed %.2f ms" % (d["value"], d["value_normalised"], c["krylov_iterations_per_newton_step"], c["ms_per_krylov_iteration"], c["ms_fixed_per_newton_step"]))
P
done
cat $L
