#!/bin/bash
# threshold control, MINC datasets, Fortran record after the ABI change
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_input.py tests/test_hip_fortran.py tests/test_hip_benchmark.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3/run14.log
cat gpurun_out/r3/run14.log
