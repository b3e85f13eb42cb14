#!/bin/bash
# round 4, pass 14: smoke(), the new multi-rank tests, and the c3 / c4 / c5 evidence sets again on the final code (the leaner
# parked records of the assembly sweeps came after the first final pass)
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2 | tee gpurun_out/r4/smoke.log
python -m pytest tests/test_hip_multirank.py -x -q -k "refused or tracer or shardings" 2>&1 | grep -v amdgpu | tail -3 | tee gpurun_out/r4/pytest_14.log
SKIP_TESTS=1 SKIP_SHARES=1 SKIP_LOOPBACK=1 bash tools/r4_final.sh
