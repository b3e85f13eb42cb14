#!/bin/bash
# round-3 GPU pass 3: new tests, then A/B runs (brick order, one-wave-per-brick kernel, ILU(k) table)
mkdir -p gpurun_out/r3
summ() { python - "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d["config"]
    print("%-34s value %.3f acc %s its/step %.1f ms/it %.4f fixed %.1f fused %.4f frac %.3f spmv %.3f %s" % (sys.argv[1].split("/")[-1], d["value"], ("%.2f" % d["value_accepted_steps"]) if d.get("value_accepted_steps") else None, c["krylov_iterations_per_newton_step"], c["ms_per_krylov_iteration"], c["ms_fixed_per_newton_step"], d["roofline"]["ms_per_launch"], d["roofline"]["frac"], d["roofline"]["spmv_frac"], d["roofline"]["kernel"][:24]))
except Exception as e:
    print(sys.argv[1], "no result:", e)
PY
}
python -m pytest tests/test_hip_pc.py tests/test_hip_multirank.py tests/test_hip_parity.py tests/test_hip_tracer.py -m gpu -x -q -s --durations=8 > gpurun_out/r3/pytest_new.log 2>&1
echo "pytest rc $?"; tail -14 gpurun_out/r3/pytest_new.log | cut -c1-220
run() { name=$1; shift; python bench.py "$@" --no-cpu > gpurun_out/r3/$name.json 2> gpurun_out/r3/$name.log; summ gpurun_out/r3/$name.json; }
run ab_c3_z --steps 20 --warmup 5
run ab_c3_x --steps 20 --warmup 5 --brick-order x
run ab_c4_rows --config c4 --steps 20 --warmup 5
run ab_c4_wave --config c4 --steps 20 --warmup 5 --brick 8 4 2
run ab_c5 --config c5 --steps 20 --warmup 5
run ilu0_c2 --config c2 --steps 20 --warmup 5
run ilu1_c2 --config c2 --steps 20 --warmup 5 --ilu-levels 1
run ilu2_c2 --config c2 --steps 20 --warmup 5 --ilu-levels 2
run asm_c2 --config c2 --steps 20 --warmup 5 --pc asm
run share8_nt --rank-share 8 --steps 20 --warmup 5
WAI_EXTRA_HIPCC_FLAGS="-DWAI_NO_NT=1" python -m waiwera_amd.build --force > /dev/null 2>&1
run share8_nont --rank-share 8 --steps 20 --warmup 5
python -m waiwera_amd.build --force > /dev/null 2>&1
