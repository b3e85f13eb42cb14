#!/bin/bash
# round 4, pass 12: ILU(0) with off-diagonal fill (a cell graph with triangles) on the default build; the fallback-kernel
# build through the pc / parity / tracer / salt tests
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
python -m pytest tests/test_hip_pc.py -q -x -k "off_diagonal_fill or fused_iteration or four_launches" 2>&1 | grep -v amdgpu | tail -12 | cut -c1-220 | tee gpurun_out/r4/pytest_12.log
bash tools/ci_fallback_kernels.sh 2>&1 | cut -c1-220 | tee gpurun_out/r4/ci_fallback_kernels.log
