#!/bin/bash
# round 4, pass 8: the first generation of k_pc_park started in cohorts (WAI_PC_STAGGER: 10-ns ticks between cohorts)
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
for st in 0 300 600 900 0; do
  echo "== WAI_PC_STAGGER=$st" | tee -a gpurun_out/r4/stagger.log
  WAI_PC_STAGGER=$st python bench.py --micro-only --rank-share 8 --spmv-reps 200 2>&1 | grep "^micro" | sed 's/^micro c3/micro c3s8/' | cut -c1-330 | tee -a gpurun_out/r4/stagger.log
  WAI_PC_STAGGER=$st python bench.py --micro-only --config c2 --spmv-reps 200 2>&1 | grep "^micro" | cut -c1-330 | tee -a gpurun_out/r4/stagger.log
done
for st in 0 600; do
  echo "== WAI_PC_STAGGER=$st" | tee -a gpurun_out/r4/stagger.log
  WAI_PC_STAGGER=$st python bench.py --micro-only --config c3 --spmv-reps 100 2>&1 | grep "^micro" | cut -c1-330 | tee -a gpurun_out/r4/stagger.log
done
