#!/bin/bash
# round 4, pass 11: wave sums by DPP row operations against the shuffle tree (v_dpp / v_shfl builds), tests first
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
python -m pytest tests/test_hip_pc.py tests/test_hip_parity.py tests/test_hip_tracer.py -x -q 2>&1 | grep -v amdgpu | tail -4 | tee gpurun_out/r4/pytest_11.log
L=gpurun_out/r4/dpp_ab.log
cp waiwera_amd/libwaiwera_hip.so /tmp/lib_keep.so
for rep in 1 2; do for v in v_shfl v_dpp; do
  cp waiwera_amd/$v.so waiwera_amd/libwaiwera_hip.so
  for cfg in c3 c4 c5; do
    python bench.py --micro-only --config $cfg --spmv-reps 100 2>&1 | grep '^micro' | sed "s/^micro/$v/" | cut -c1-400 | tee -a $L
  done
  python bench.py --micro-only --rank-share 8 --spmv-reps 100 2>&1 | grep '^micro' | sed "s/^micro c3/$v c3s8/" | cut -c1-400 | tee -a $L
done; done
cp /tmp/lib_keep.so waiwera_amd/libwaiwera_hip.so
