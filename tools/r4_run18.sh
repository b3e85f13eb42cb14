#!/bin/bash
# round 4, pass 18: k_pc_wave<HELP> (a long row's trailing slots streamed by a short row's lane) against one lane per row
# (WAI_WAVE_HELP=0), C5, alternating on one box; the tests that run MINC bricks first
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -k "minc or MINC or c5 or wave or shard or 3x3" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 | tee gpurun_out/r4/pytest_18.log
L=gpurun_out/r4/wave_help_ab.log
for rep in 1 2 3; do for h in 0 1; do
  WAI_WAVE_HELP=$h python bench.py --micro-only --config c5 --spmv-reps 100 2>&1 | grep '^micro' | sed "s/^micro/help=$h/" | cut -c1-420 | tee -a $L
done; done
for h in 0 1; do
  WAI_WAVE_HELP=$h python bench.py --config c5 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r4/bench_c5_help$h.json
done
