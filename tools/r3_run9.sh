#!/bin/bash
mkdir -p gpurun_out/r3
micro() { python bench.py "$@" --micro-only 2>&1 | grep "^micro" | cut -c1-200; }
line() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); c=d["config"]
print(sys.argv[1].split("/")[-1], "value %.3f its/step %.1f ms/it %.4f dev-only %s fused %.4f vectors %s fixed %.2f" % (d["value"], c["krylov_iterations_per_newton_step"], c["ms_per_krylov_iteration"], c.get("ms_per_krylov_iteration_device_only"), d["roofline"]["ms_per_launch"], c.get("ms_vector_updates_per_iteration"), c["ms_fixed_per_newton_step"]))
PY
}
python -m pytest tests/test_hip_parity.py tests/test_hip_pc.py -m gpu -x -q 2>&1 | tail -2
echo "== default build (16-byte gathers, longest-first launch order)"
micro --config c4; micro --config c5; micro --rank-share 8; micro
python bench.py --rank-share 8 --steps 20 --warmup 5 --no-cpu > gpurun_out/r3/b9_share8.json 2> gpurun_out/r3/b9_share8.log; line gpurun_out/r3/b9_share8.json
python bench.py --config c4 --steps 20 --warmup 5 --no-cpu > gpurun_out/r3/b9_c4.json 2> gpurun_out/r3/b9_c4.log; line gpurun_out/r3/b9_c4.json
WAI_EXTRA_HIPCC_FLAGS="-DWAI_X_SCALAR_GATHER" python -m waiwera_amd.build --force > /dev/null 2>&1
echo "== -DWAI_X_SCALAR_GATHER"
micro --config c4; micro --config c5
python -m waiwera_amd.build --force > /dev/null 2>&1
