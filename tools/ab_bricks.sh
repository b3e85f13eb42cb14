# A/B of the preconditioner brick shape on the 216^3 workload (1 x MI355X).  No warm-up and only the
# first Newton iterations of the first time step, so every shape solves the same linear systems:
# Krylov iterations of Newton iterations 1-3, fused-kernel time, and their product.
for b in "8 8 8" "12 12 3" "18 12 2" "18 18 1" "27 18 1" "36 12 1" "24 18 1" "18 9 3" "27 9 2" "12 12 2" "9 9 6" "12 9 4" "54 9 1" "108 4 1"; do
  python bench.py --brick $b --steps 4 --warmup 0 --no-cpu 2>&1 | grep -E "^\{|  step " | python -c "
import sys, json
its = []
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line)
        ms = d['roofline']['ms_per_launch']
    else:
        p = line.split()
        its.append(int(p[7]))
k3 = sum(its[:3])
print('brick %-10s krylov its %s  sum(1-3) %5d  pc ms %.3f  -> %.2f s of k_pc for the three solves' % ('$b', its[:4], k3, ms, 2 * k3 * ms * 1e-3))
"
done
