#!/bin/bash
# round 4, pass 24: 3 x 3 / 4 x 4 values with the elements of 64 rows together inside a slot (v_slot) against one plane per
# element (v_planes), C4, C5 and C4's four-rank share, alternating on one box; the whole GPU suite first on the new layout
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee gpurun_out/r4/pytest_24.log
L=gpurun_out/r4/slot_layout_ab.log
cp waiwera_amd/libwaiwera_hip.so /tmp/lib_keep.so
for rep in 1 2 3; do for v in v_planes v_slot; do
  cp waiwera_amd/$v.so waiwera_amd/libwaiwera_hip.so
  for cfg in c4 c5; do
    python bench.py --micro-only --config $cfg --spmv-reps 100 2>&1 | grep '^micro' | sed "s/^micro/$v/" | cut -c1-420 | tee -a $L
  done
  python bench.py --micro-only --config c4 --rank-share 4 --spmv-reps 100 2>&1 | grep '^micro' | sed "s/^micro c4/$v c4s4/" | cut -c1-420 | tee -a $L
done; done
cp /tmp/lib_keep.so waiwera_amd/libwaiwera_hip.so
