#!/bin/bash
# round 4, pass 9: cohort delay of the first generation -- k_pc_park around its default, k_pc_wave (C4, C5 and their shares)
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
L=gpurun_out/r4/stagger2.log
m() { tag=$1; shift; python bench.py --micro-only --spmv-reps 200 "$@" 2>&1 | grep "^micro" | grep -v "iteration device" | sed "s/^micro [a-z0-9]*/micro $tag/" | cut -c1-300 | tee -a $L; }
for st in 0 400 600 800 0 600; do
  echo "== WAI_PC_STAGGER=$st" | tee -a $L
  export WAI_PC_STAGGER=$st
  m c3s8 --rank-share 8
  m c3 --config c3
done
for st in 0 200 400 600 0 400; do
  echo "== WAI_PC_STAGGER=$st" | tee -a $L
  export WAI_PC_STAGGER=$st
  m c4 --config c4
  m c5 --config c5
  m c4s4 --config c4 --rank-share 4
  m c5s2 --config c5 --rank-share 2
done
unset WAI_PC_STAGGER
python -m pytest tests/test_hip_pc.py tests/test_hip_parity.py -x -q 2>&1 | grep -v amdgpu | tail -3 | tee gpurun_out/r4/pytest_9.log
