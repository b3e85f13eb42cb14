#!/bin/bash
# round 4, pass 26: k_pc_wave's finalisers taking their virtual threads two at a time (v_jv2) against one at a time (v_jv1),
# C4 / C5, alternating on one box; pc / parity / multirank tests on the new build first (same bits expected)
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
python -m pytest tests/test_hip_pc.py tests/test_hip_parity.py tests/test_hip_multirank.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee gpurun_out/r4/pytest_26.log
L=gpurun_out/r4/fin_jv_ab.log
cp waiwera_amd/libwaiwera_hip.so /tmp/lib_keep.so
for rep in 1 2 3; do for v in v_jv1 v_jv2; do
  cp waiwera_amd/$v.so waiwera_amd/libwaiwera_hip.so
  for cfg in c4 c5; do
    python bench.py --micro-only --config $cfg --spmv-reps 100 2>&1 | grep '^micro' | sed "s/^micro/$v/" | cut -c1-420 | tee -a $L
  done
done; done
cp /tmp/lib_keep.so waiwera_amd/libwaiwera_hip.so
