#!/bin/bash
# network couplings across ranks
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3
timeout 1500 python -m pytest tests/test_hip_multirank.py tests/test_hip_input.py -m gpu -x -q -s -k "network or coupling or reinjection or makeup" 2>&1 | tail -25 > gpurun_out/r3/run15.log
cat gpurun_out/r3/run15.log
