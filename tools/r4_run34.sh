#!/bin/bash
# round 4, pass 34: C3 with balanced bricks (14 x 14 bricks of 15-16 x 15-16 x 2 per layer pair) against full bricks + remainders
# (13 of 16 and one of 8 per axis): the kernel microbenchmarks, alternating, and one bench line each (check included)
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
L=gpurun_out/r4/balanced_bricks_ab.log
for rep in 1 2 3; do for b in 0 1; do
  python bench.py --micro-only --config c3 --balanced-bricks $b --spmv-reps 100 2>&1 | grep '^micro.*\(\[k_pc\|iteration\)' | sed "s/^micro/balanced=$b/" | cut -c1-200 | tee -a $L
done; done
for b in 0 1; do
  python bench.py --config c3 --balanced-bricks $b --steps 20 --warmup 5 --no-cpu 2>/dev/null | tail -1 > gpurun_out/r4/bench_c3_balanced$b.json
  python -c "
import json; d=json.load(open('gpurun_out/r4/bench_c3_balanced$b.json')); c=d['config']
print('balanced=$b bench: value %.3f acc %.2f its/step %.1f ms/it %.4f fused %.4f frac %.3f check %s' % (d['value'], d['value_accepted_steps'], c['krylov_iterations_per_newton_step'], c['ms_per_krylov_iteration'], d['roofline']['ms_per_launch'], d['roofline']['frac'], d['check'].get('passed')))" | tee -a $L
done
