#!/bin/bash
# round 4, pass 15: next slot's block requested early in k_pc_park's slot loop (v_pref) against the default (v_base), under cohorts
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
L=gpurun_out/r4/prefetch_ab.log
cp waiwera_amd/libwaiwera_hip.so /tmp/lib_keep.so
for rep in 1 2; do for v in v_base v_pref; do
  cp waiwera_amd/$v.so waiwera_amd/libwaiwera_hip.so
  for cfg in c3 c2; do
    python bench.py --micro-only --config $cfg --spmv-reps 100 2>&1 | grep '^micro' | sed "s/^micro/$v/" | cut -c1-400 | tee -a $L
  done
  python bench.py --micro-only --rank-share 8 --spmv-reps 100 2>&1 | grep '^micro' | sed "s/^micro c3/$v c3s8/" | cut -c1-400 | tee -a $L
done; done
cp /tmp/lib_keep.so waiwera_amd/libwaiwera_hip.so
