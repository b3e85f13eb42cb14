#!/bin/bash
# round 4, pass 30: k_pc_wave -- partner vector in the epilogue (v_late), at the start of the brick (v_early: the default now), and
# the operand's own entries requested a backward sweep early on top (v_early2, -DWAI_WAVE_EARLY_XI); the full GPU suite on v_early first
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
cp waiwera_amd/v_early.so waiwera_amd/libwaiwera_hip.so; python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee gpurun_out/r4/pytest_30.log
L=gpurun_out/r4/wave_early_xi_ab.log
cp waiwera_amd/libwaiwera_hip.so /tmp/lib_keep.so
for rep in 1 2 3; do for v in v_late v_early v_early2; do
  cp waiwera_amd/$v.so waiwera_amd/libwaiwera_hip.so
  for cfg in c4 c5; do
    python bench.py --micro-only --config $cfg --spmv-reps 100 2>&1 | grep '^micro' | sed "s/^micro/$v/" | cut -c1-420 | tee -a $L
  done
  python bench.py --micro-only --config c4 --rank-share 4 --spmv-reps 100 2>&1 | grep '^micro' | sed "s/^micro c4/$v c4s4/" | cut -c1-420 | tee -a $L
done; done
cp /tmp/lib_keep.so waiwera_amd/libwaiwera_hip.so
