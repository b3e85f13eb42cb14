#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_hip_parity.py -q -m gpu -s -k "test_jacobian or residual_forms" -p no:cacheprovider 2>&1 | grep -E "arbiter audit|jacobian parity|passed|failed|FAILED|Error" > gpurun_out/arbiter_calibration_r6.log
tail -5 gpurun_out/arbiter_calibration_r6.log
python -m pytest tests/test_hip_input.py::test_preconditioner_choice_of_an_input_is_explicit tests/test_hip_multirank.py::test_a_deliverability_source_on_one_rank_only tests/test_hip_real_rccl.py tests/test_hip_pc.py tests/test_abi.py -q -m gpu -p no:cacheprovider -rs 2>&1 | tail -15 > gpurun_out/new_tests_r6.log
cat gpurun_out/new_tests_r6.log
( time python bench.py > gpurun_out/bench_r6a_c3.json 2> gpurun_out/bench_r6a_c3.log ) 2>&1 | tail -3
tail -4 gpurun_out/bench_r6a_c3.log | cut -c1-300
python bench.py --config c3 --ksp gmres --no-cpu --steps 10 --warmup 3 > gpurun_out/bench_r6a_c3_gmres.json 2> gpurun_out/bench_r6a_c3_gmres.log
tail -2 gpurun_out/bench_r6a_c3_gmres.log | cut -c1-400
bash tools/r6_valu.sh r6 c3 > /dev/null 2>&1
head -50 gpurun_out/valu_r6_c3.txt
