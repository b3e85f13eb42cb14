#!/bin/bash
# round 4, pass 25: 2 x 2 values with the two block rows of 64 rows together inside a slot (v_group2, -DWAI_ELL_GROUP2) against
# the (slot, block row) planes (v_base2), C3, C2, C3's eight-rank share, alternating on one box; tests on the variant first
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
cp waiwera_amd/libwaiwera_hip.so /tmp/lib_keep.so
cp waiwera_amd/v_group2.so waiwera_amd/libwaiwera_hip.so
python -m pytest tests/test_hip_pc.py tests/test_hip_parity.py tests/test_hip_tracer.py tests/test_hip_multirank.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee gpurun_out/r4/pytest_25.log
L=gpurun_out/r4/group2_ab.log
for rep in 1 2 3; do for v in v_base2 v_group2; do
  cp waiwera_amd/$v.so waiwera_amd/libwaiwera_hip.so
  for cfg in c3 c2; do
    python bench.py --micro-only --config $cfg --spmv-reps 100 2>&1 | grep '^micro' | sed "s/^micro/$v/" | cut -c1-420 | tee -a $L
  done
  python bench.py --micro-only --rank-share 8 --spmv-reps 100 2>&1 | grep '^micro' | sed "s/^micro c3/$v c3s8/" | cut -c1-420 | tee -a $L
done; done
cp /tmp/lib_keep.so waiwera_amd/libwaiwera_hip.so
