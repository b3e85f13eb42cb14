# A/B of the preconditioner brick shape on the 216^3 workload (1 x MI355X)
for b in "8 8 8" "9 9 6" "6 6 6" "9 9 9" "12 9 9"; do
  echo "brick $b"
  python bench.py --brick $b --steps 20 --warmup 10 --no-cpu 2>&1 | grep -E "fused pc|^\{" | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line)
        print('  value %.3f steps/s  ms/step %.0f  krylov/newton %.0f  pc frac %.3f' % (d['value'], d['ms_per_step'], d['config']['krylov_iterations_per_newton_step'], d['roofline']['frac']))
    else:
        print('  ' + line.strip())
"
done
