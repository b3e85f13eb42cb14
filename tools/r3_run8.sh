#!/bin/bash
mkdir -p gpurun_out/r3
python -m pytest tests/test_hip_parity.py tests/test_hip_pc.py tests/test_hip_multirank.py tests/test_hip_input.py -m gpu -x -q 2>&1 | tail -3
line() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); c=d["config"]
print(sys.argv[1].split("/")[-1], "value %.3f its/step %.1f ms/it %.4f dev-only %s fused %.4f vectors %s fixed %.2f" % (d["value"], c["krylov_iterations_per_newton_step"], c["ms_per_krylov_iteration"], c.get("ms_per_krylov_iteration_device_only"), d["roofline"]["ms_per_launch"], c.get("ms_vector_updates_per_iteration"), c["ms_fixed_per_newton_step"]))
PY
}
for i in 1 2; do python bench.py --rank-share 8 --steps 20 --warmup 5 --no-cpu > gpurun_out/r3/b8_share8_$i.json 2> gpurun_out/r3/b8_share8_$i.log; line gpurun_out/r3/b8_share8_$i.json; done
python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/r3/b8_c3.json 2> gpurun_out/r3/b8_c3.log; line gpurun_out/r3/b8_c3.json
