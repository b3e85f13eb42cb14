#!/bin/bash
# round 4, pass 29: k_pc_wave requesting the dot product's partner vector at the start of the brick (v_early, -DWAI_WAVE_EARLY_AUX)
# against in the epilogue (v_late), C4 / C5 / C4's share, alternating on one box
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
cp waiwera_amd/v_early.so waiwera_amd/libwaiwera_hip.so; python -m pytest tests/test_hip_pc.py tests/test_hip_parity.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee gpurun_out/r4/pytest_29.log
L=gpurun_out/r4/wave_early_aux_ab.log
cp waiwera_amd/libwaiwera_hip.so /tmp/lib_keep.so
for rep in 1 2 3; do for v in v_late v_early; do
  cp waiwera_amd/$v.so waiwera_amd/libwaiwera_hip.so
  for cfg in c4 c5; do
    python bench.py --micro-only --config $cfg --spmv-reps 100 2>&1 | grep '^micro' | sed "s/^micro/$v/" | cut -c1-420 | tee -a $L
  done
  python bench.py --micro-only --config c4 --rank-share 4 --spmv-reps 100 2>&1 | grep '^micro' | sed "s/^micro c4/$v c4s4/" | cut -c1-420 | tee -a $L
done; done
cp /tmp/lib_keep.so waiwera_amd/libwaiwera_hip.so
