#!/bin/bash
# round 4, pass 27: the finalisers' polling interval -- s_sleep 8 (v_jv1 = HEAD), 32, 100 -- C4 / C5, alternating on one box
mkdir -p gpurun_out/r4
export TMPDIR=/tmp

L=gpurun_out/r4/fin_sleep_ab.log
cp waiwera_amd/libwaiwera_hip.so /tmp/lib_keep.so
for rep in 1 2 3; do for v in v_jv1 v_sl32 v_sl100; do
  cp waiwera_amd/$v.so waiwera_amd/libwaiwera_hip.so
  for cfg in c4 c5; do
    python bench.py --micro-only --config $cfg --spmv-reps 100 2>&1 | grep '^micro' | sed "s/^micro/$v/" | cut -c1-420 | tee -a $L
  done
done; done
cp /tmp/lib_keep.so waiwera_amd/libwaiwera_hip.so
