#!/bin/bash
# one bench line per option set on one box.  usage: bash tools/opt_scan.sh <config> "<extra bench args>" ...
CFG=$1; shift
for o in "$@"; do
  echo "$CFG [$o]: $(timeout 400 python bench.py --config $CFG --steps ${STEPS:-20} --warmup ${WARM:-5} --no-cpu $o 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],3), 'steps/s; its/step', round(d['config']['krylov_iterations_per_newton_step'],1), 'ms/it', round(d['config']['ms_per_krylov_iteration'],4), 'fused', round(d['roofline']['frac'],3), 'spmv', round(d['roofline']['spmv_frac'],3))" 2>&1 | tail -1)"
done
