#!/bin/bash
# round 4, pass 31: the cohort delay of k_pc_park's first generation (WAI_PC_STAGGER, 10-ns ticks; default 600) scanned again on
# the round's final kernels, C3 and its eight-rank share, one box
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
L=gpurun_out/r4/stagger_scan_final.log
for rep in 1 2; do for t in 0 300 600 900 1200; do
  WAI_PC_STAGGER=$t python bench.py --micro-only --config c3 --spmv-reps 100 2>&1 | grep '^micro.*\(\[k_pc\|iteration\)' | sed "s/^micro/ticks=$t/" | cut -c1-200 | tee -a $L
  WAI_PC_STAGGER=$t python bench.py --micro-only --rank-share 8 --spmv-reps 100 2>&1 | grep '^micro.*\(\[k_pc\|iteration\)' | sed "s/^micro c3/ticks=$t c3s8/" | cut -c1-200 | tee -a $L
done; done
