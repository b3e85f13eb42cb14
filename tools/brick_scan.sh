# Newton steps/s, Krylov iterations and fused-kernel roofline fraction against brick shape
# usage: bash tools/brick_scan.sh <config> "<bx by bz>" ...
CFG=$1; shift
for b in "$@"; do
  echo "$CFG brick $b: $(timeout 400 python bench.py --config $CFG --steps ${STEPS:-10} --warmup ${WARM:-2} --no-cpu --brick $b 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],3), 'steps/s; its/step', round(d['config']['krylov_iterations_per_newton_step'],1), 'ms/it', round(d['config']['ms_per_krylov_iteration'],3), 'fused', round(d['roofline']['frac'],3))" 2>&1 | tail -1)"
done
