#!/bin/bash
# round 4, pass 21: k_pc_wave<WC = 7> (seven slots known at compile time: the slots' loads in flight together) against the
# general slot loop (WAI_WAVE_PIPE=0), C4, alternating on one box; 3 x 3 tests first
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
python -m pytest tests/test_hip_pc.py tests/test_hip_parity.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee gpurun_out/r4/pytest_21.log
L=gpurun_out/r4/wave_pipe_ab.log
for rep in 1 2 3; do for h in 0 1; do
  WAI_WAVE_PIPE=$h python bench.py --micro-only --config c4 --spmv-reps 100 2>&1 | grep '^micro' | sed "s/^micro/pipe=$h/" | cut -c1-420 | tee -a $L
done; done
