#!/bin/bash
# round-3 GPU pass 5: the whole GPU suite, then the evidence runs (bench line + rocprofv3 kernel trace + PMC passes)
mkdir -p gpurun_out/r3
python -m pytest tests -m gpu -q -s --durations=12 > gpurun_out/r3/pytest_gpu_full.log 2>&1
echo "pytest rc $?"; tail -18 gpurun_out/r3/pytest_gpu_full.log | cut -c1-200
set +x
for cfg in c3 c4 c5; do
  bash tools/gpu_profile.sh r3 $cfg --steps 20 --warmup 5 > gpurun_out/r3/gpu_profile_$cfg.log 2>&1
  head -8 gpurun_out/rocprof_kernel_stats_r3_$cfg.txt | cut -c1-140
  python -c "
import json
d=json.load(open('gpurun_out/bench_r3_$cfg.json')); c=d['config']
print('$cfg value %.3f acc %s its/step %.1f ms/it %.4f frac %.3f spmv %.3f check %s' % (d['value'], d.get('value_accepted_steps'), c['krylov_iterations_per_newton_step'], c['ms_per_krylov_iteration'], d['roofline']['frac'], d['roofline']['spmv_frac'], d['check'].get('passed')))
print(json.load(open('gpurun_out/pmc_traffic_r3_$cfg.json')).get('k_pc_traffic_over_algorithmic'))
" 2>&1 | tail -3
done
