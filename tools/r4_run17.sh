#!/bin/bash
# round 4, pass 17: k_pc_wave with the epilogue's dot-product partners requested before the backward sweep (v_wpre) against
# the same code without (v_b1024), C4 / C5, alternating on one box; tests of the 3 x 3 kernels first
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
python -m pytest tests/test_hip_pc.py tests/test_hip_parity.py tests/test_hip_tracer.py tests/test_hip_multirank.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee gpurun_out/r4/pytest_17.log
L=gpurun_out/r4/wave_prefetch_ab.log
cp waiwera_amd/libwaiwera_hip.so /tmp/lib_keep.so
for rep in 1 2 3; do for v in v_b1024 v_wpre; do
  cp waiwera_amd/$v.so waiwera_amd/libwaiwera_hip.so
  for cfg in c4 c5; do
    python bench.py --micro-only --config $cfg --spmv-reps 100 2>&1 | grep '^micro' | sed "s/^micro/$v/" | cut -c1-420 | tee -a $L
  done
done; done
cp /tmp/lib_keep.so waiwera_amd/libwaiwera_hip.so
rocm-smi --showclocks --showtemp --showpower --json > gpurun_out/r4/rocm_smi_sample.json 2>&1
