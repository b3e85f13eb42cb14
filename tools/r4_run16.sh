#!/bin/bash
# round 4, pass 16: finaliser variants on one box -- v_old (HEAD before: entries and slots polled one after the other),
# v_b256 / v_b512 / v_b1024 (everything not yet arrived asked for again together, once per round; slice size)
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
python -m pytest tests/test_hip_pc.py tests/test_hip_parity.py tests/test_hip_tracer.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee gpurun_out/r4/pytest_16.log
L=gpurun_out/r4/fin_ab2.log
cp waiwera_amd/libwaiwera_hip.so /tmp/lib_keep.so
for rep in 1 2; do for v in v_old v_b256 v_b512 v_b1024; do
  cp waiwera_amd/$v.so waiwera_amd/libwaiwera_hip.so
  python bench.py --micro-only --rank-share 8 --spmv-reps 100 2>&1 | grep '^micro' | sed "s/^micro c3/$v c3s8/" | cut -c1-420 | tee -a $L
  for cfg in c3 c4 c5; do
    python bench.py --micro-only --config $cfg --spmv-reps 100 2>&1 | grep '^micro' | sed "s/^micro/$v/" | cut -c1-420 | tee -a $L
  done
done; done
cp /tmp/lib_keep.so waiwera_amd/libwaiwera_hip.so
