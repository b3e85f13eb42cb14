#!/bin/bash
# round 4, pass 22: k_spmv with the branch-free seven-slot row loop (v_w7) against the general loop (v_now7), alternating
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
L=gpurun_out/r4/spmv_w7_ab.log
cp waiwera_amd/libwaiwera_hip.so /tmp/lib_keep.so
for rep in 1 2 3; do for v in v_now7 v_w7; do
  cp waiwera_amd/$v.so waiwera_amd/libwaiwera_hip.so
  for cfg in c3 c4 c2; do
    python bench.py --micro-only --config $cfg --spmv-reps 200 2>&1 | grep '^micro.*\[' | sed "s/^micro/$v/" | cut -c1-200 | tee -a $L
  done
done; done
cp /tmp/lib_keep.so waiwera_amd/libwaiwera_hip.so
python -m pytest tests/test_hip_pc.py tests/test_hip_parity.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee gpurun_out/r4/pytest_22.log
