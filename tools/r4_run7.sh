#!/bin/bash
# round 4, pass 7: the reference's own layout -- ONE ILU(0) block per rank on the launch-per-level path -- at C2 and C3;
# assembly sweeps at C5 with the cells sorted by face count; parity test of the sorted sweeps
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
python -m pytest tests/test_hip_parity.py -x -q -k "sorted_by_face_count or minc or kernels_agree" 2>&1 | grep -v amdgpu | tail -4 | cut -c1-200 | tee gpurun_out/r4/pytest_7.log
python bench.py --config c2 --pc ilu --no-cpu --steps 10 --warmup 2 > gpurun_out/r4/bench_r4_c2_ilu_one_block.json 2> gpurun_out/r4/bench_r4_c2_ilu_one_block.log
python bench.py --config c2 --no-cpu --steps 10 --warmup 2 > gpurun_out/r4/bench_r4_c2_bricks_10steps.json 2> gpurun_out/r4/bench_r4_c2_bricks_10steps.log
timeout 900 python bench.py --config c3 --pc ilu --no-cpu --steps 6 --warmup 1 > gpurun_out/r4/bench_r4_c3_ilu_one_block.json 2> gpurun_out/r4/bench_r4_c3_ilu_one_block.log
python - <<'PY'
import json
for n in ["bench_r4_c2_ilu_one_block","bench_r4_c2_bricks_10steps","bench_r4_c3_ilu_one_block"]:
    try:
        d=json.load(open("gpurun_out/r4/%s.json" % n)); c=d["config"]
        print("%-28s value %.3f acc %.2f its/step %.1f ms/it %.4f fixed %.2f launches/it %.1f kernel %s" % (n, d["value"], d.get("value_accepted_steps") or 0, c["krylov_iterations_per_newton_step"], c["ms_per_krylov_iteration"], c["ms_fixed_per_newton_step"], c["launches_per_krylov_iteration"], d["roofline"]["kernel"][:50]))
    except Exception as e: print(n, e)
PY
bash tools/asm_traffic.sh r4b c5
