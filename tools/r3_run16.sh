#!/bin/bash
# full GPU suite on the code with cross-rank network couplings
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3
timeout 2000 python -m pytest tests -m gpu -x -q -s --durations=5 > gpurun_out/r3/pytest_gpu_last.log 2>&1
echo "pytest rc $?"
tail -8 gpurun_out/r3/pytest_gpu_last.log
