# A/B: k_pc_park with two parked upper blocks per row (40 KB LDS, 4 workgroups per CU) against three
WAI_PC_PARK2=1 timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -k "spmv_ilu_krylov or timesteps" 2>&1 | grep -E "passed|failed|FAILED" | tail -3
for v in 0 1; do
  if [ $v = 1 ]; then export WAI_PC_PARK2=1; else unset WAI_PC_PARK2; fi
  echo "park2=$v"
  python bench.py --steps 8 --warmup 4 --no-cpu --spmv-reps 50 2>&1 | grep -E "spmv:|^\{" | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('  steps/s %.3f  ms/step %.1f  krylov/newton %.0f' % (d['value'], d['ms_per_step'], d['config']['krylov_iterations_per_newton_step']))
    else: print(' ', line.strip()[:200])
"
done
