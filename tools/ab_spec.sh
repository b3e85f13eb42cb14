# A/B: speculative first half of the next BiCGStab iteration (hides the residual-norm read-back)
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_hip_tracer.py tests/test_hip_multirank.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED" | tail -3
for v in 1 0; do
  if [ $v = 1 ]; then export WAI_BCGS_NO_SPECULATION=1; else unset WAI_BCGS_NO_SPECULATION; fi
  echo "no_speculation=$v"
  python bench.py --steps 8 --warmup 4 --no-cpu --spmv-reps 20 2>&1 | grep -E "^\{" | python -c "
import sys, json
for line in sys.stdin:
    d = json.loads(line); print('  steps/s %.3f  ms/step %.1f  krylov/newton %.0f' % (d['value'], d['ms_per_step'], d['config']['krylov_iterations_per_newton_step']))
"
done
