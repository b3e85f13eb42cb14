# the composed (S = R - alpha V formed inside the second fused launch) against the stored-S BiCGStab iteration END TO END:
# full bench lines, alternating, one box.   bash tools/ab_compose_full.sh <log name> "<bench args>" ...
NAME=${1:-compose_full_r5}; shift
[ $# -eq 0 ] && set -- "--config c3" "--rank-share 8" "--config c2"
mkdir -p gpurun_out
: > gpurun_out/$NAME.log
for r in 1 2; do
for v in 0 1; do
  for cfg in "$@"; do
    echo "== round $r COMPOSE=$v $cfg" >> gpurun_out/$NAME.log
    WAI_BCGS_COMPOSE=$v python bench.py $cfg --steps 12 --warmup 3 --no-cpu --spmv-reps 50 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d['config']
print('value %.3f norm %s its/step %.1f ms/it %.4f fixed %.2f dev-only %s' % (d['value'], d.get('value_normalised'), c['krylov_iterations_per_newton_step'], c['ms_per_krylov_iteration'], c['ms_fixed_per_newton_step'], c.get('ms_per_krylov_iteration_device_only')))" >> gpurun_out/$NAME.log
  done
done
done
cat gpurun_out/$NAME.log
