#!/bin/bash
# round 4, pass 6: static LDS back on a 16-byte multiple (k_pc_park's b128 LDS accesses), finaliser slices of 1024; the
# whole GPU suite on the split library
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
for cfg in c3 c2 c4 c5; do
  python bench.py --micro-only --config $cfg --spmv-reps 100 2>&1 | grep "^micro" | tee -a gpurun_out/r4/micro_modes6.log
done
python bench.py --micro-only --rank-share 8 --spmv-reps 100 2>&1 | grep "^micro" | sed 's/^micro c3/micro c3s8/' | tee -a gpurun_out/r4/micro_modes6.log
python -m pytest tests -m gpu -q --durations=8 2>&1 | grep -v amdgpu | tail -25 | cut -c1-220 | tee gpurun_out/r4/pytest_gpu6.log
