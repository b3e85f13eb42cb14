# Whole-protocol A/B of preconditioner brick shapes (default bench: 10 warm-up + 20 timed Newton
# steps at 216^3); the transient is chaotic in its step failures, so repeat shapes to see the spread
shapes=("8 8 8" "16 16 2" "20 12 2" "18 12 2" "12 18 2" "16 16 1" "32 16 1" "24 20 1" "12 12 3" "16 8 4" "8 8 4" "16 16 3")
for b in "${shapes[@]}"; do
  python bench.py --brick $b --steps 20 --warmup 10 --no-cpu 2>&1 | grep -E "^\{" | python -c "
import sys, json
for line in sys.stdin:
    d = json.loads(line)
    print('brick %-9s value %.3f steps/s  ms/step %.0f  krylov/newton %.0f  pc ms %.3f frac %.3f' % ('$b', d['value'], d['ms_per_step'], d['config']['krylov_iterations_per_newton_step'], d['roofline']['ms_per_launch'], d['roofline']['frac']))
"
done
