#!/bin/bash
# round 4, pass 19: k_pc_wave with two bricks per workgroup (14 bricks per CU at C4 instead of 12) against four, C4 and C5,
# alternating on one box
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
python -m pytest tests/test_hip_pc.py tests/test_hip_parity.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee gpurun_out/r4/pytest_19.log
L=gpurun_out/r4/wave_bpw_ab.log
for rep in 1 2 3; do for b in 4 2; do
  for cfg in c4 c5; do
    WAI_WAVE_BPW=$b python bench.py --micro-only --config $cfg --spmv-reps 100 2>&1 | grep '^micro' | sed "s/^micro/bpw=$b/" | cut -c1-420 | tee -a $L
  done
done; done
