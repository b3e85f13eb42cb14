#!/bin/bash
# round 4, pass 28: k_pc_wave storing one partial sum per workgroup (v_perwg) against one per brick (v_perbrick), C4 / C5 and
# C4's four-rank share, alternating on one box; the whole GPU suite on the new build first
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee gpurun_out/r4/pytest_28.log
L=gpurun_out/r4/wave_partials_ab.log
cp waiwera_amd/libwaiwera_hip.so /tmp/lib_keep.so
for rep in 1 2 3; do for v in v_perbrick v_perwg; do
  cp waiwera_amd/$v.so waiwera_amd/libwaiwera_hip.so
  for cfg in c4 c5; do
    python bench.py --micro-only --config $cfg --spmv-reps 100 2>&1 | grep '^micro' | sed "s/^micro/$v/" | cut -c1-420 | tee -a $L
  done
  python bench.py --micro-only --config c4 --rank-share 4 --spmv-reps 100 2>&1 | grep '^micro' | sed "s/^micro c4/$v c4s4/" | cut -c1-420 | tee -a $L
done; done
cp /tmp/lib_keep.so waiwera_amd/libwaiwera_hip.so
