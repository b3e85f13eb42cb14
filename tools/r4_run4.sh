#!/bin/bash
# round 4, pass 4: what the reductions cost the fused launch, by mode and config (micro only); padded-vector SpMV micro
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
for cfg in c3 c4 c5 c2; do
  python bench.py --micro-only --config $cfg --spmv-reps 100 2>&1 | grep "^micro" | tee -a gpurun_out/r4/micro_modes.log
done
python bench.py --micro-only --rank-share 8 --spmv-reps 100 2>&1 | grep "^micro" | sed 's/^micro c3/micro c3s8/' | tee -a gpurun_out/r4/micro_modes.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/spmv3_variants.hip -o /tmp/spmv3_variants 2>/dev/null
for b in "8 4 2" "8 5 2"; do echo "bricks $b"; /tmp/spmv3_variants $b; done 2>&1 | tee gpurun_out/r4/spmv3_variants.log
python -m pytest tests/test_hip_multirank.py -x -q -k "tracer or shardings" 2>&1 | grep -v amdgpu | tail -8 | cut -c1-200 | tee gpurun_out/r4/pytest_multirank4.log
