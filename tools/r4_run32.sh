#!/bin/bash
# round 4, pass 32: WAI_PC_STAGGER at the full C3 size again, finer and on another box: 200 / 300 / 400 / 600, three runs each
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
L=gpurun_out/r4/stagger_scan_c3.log
for rep in 1 2 3; do for t in 600 300 200 400; do
  WAI_PC_STAGGER=$t python bench.py --micro-only --config c3 --spmv-reps 100 2>&1 | grep '^micro.*\(\[k_pc\|iteration\)' | sed "s/^micro/ticks=$t/" | cut -c1-200 | tee -a $L
done; done
