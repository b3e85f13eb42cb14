#!/bin/bash
# round 4, pass 20: k_pc_wave taking a MINC row's slot count from the brick's record (default) against the row pointers
# (WAI_WAVE_ROWPTR=1: one more dependent round trip before the first block load), C5, alternating on one box
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -k "minc or MINC or c5 or wave or shard or 3x3 or pc" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 | tee gpurun_out/r4/pytest_20.log
L=gpurun_out/r4/wave_cnt_ab.log
for rep in 1 2 3; do for h in 1 0; do
  if [ $h = 1 ]; then export WAI_WAVE_ROWPTR=1; else unset WAI_WAVE_ROWPTR; fi
  python bench.py --micro-only --config c5 --spmv-reps 100 2>&1 | grep '^micro' | sed "s/^micro/rowptr=$h/" | cut -c1-420 | tee -a $L
done; done
