#!/bin/bash
# round 4, pass 13: leaner parked records (no internal energies, no permeability factor outside the salt EOS) and max_deg
# base-term slots: four 64-thread workgroups of k_jacobian_park per CU for 3 x 3 blocks -- identity tests, durations
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
python -m pytest tests/test_hip_parity.py tests/test_hip_salt.py -x -q 2>&1 | grep -v amdgpu | tail -4 | tee gpurun_out/r4/pytest_13.log
bash tools/kernel_time.sh "k_jacobian|k_residual|k_eos" c4 c5 c3 2>&1 | tee gpurun_out/r4/asm_times_lean_park.log
