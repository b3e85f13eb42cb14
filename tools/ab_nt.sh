# A/B: streaming hints / two elements per lane in the Krylov kernels, whole Newton steps
for flags in "-DWAI_NO_NT=1" "" "-DWAI_VEC2=1"; do
  WAI_EXTRA_HIPCC_FLAGS="$flags" python -m waiwera_amd.build --force > /dev/null 2>&1
  echo "flags: '$flags'"
  python bench.py --steps 8 --warmup 4 --no-cpu --spmv-reps 50 2>&1 | grep -E "spmv:|^\{" | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('  steps/s %.3f  ms/step %.1f  krylov/newton %.0f' % (d['value'], d['ms_per_step'], d['config']['krylov_iterations_per_newton_step']))
    else: print(' ', line.strip()[:200])
"
done
python -m waiwera_amd.build --force > /dev/null 2>&1
