#!/bin/bash
# pivot recurrence for one-wave bricks: A_ik in registers (lib_piv1) against the LDS-staged kernel (lib_piv0)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
cp waiwera_amd/libwaiwera_hip.so /tmp/lib_keep.so
out=$PWD/gpurun_out/r3/run21.log; : > $out
for v in lib_piv0 lib_piv1; do
  cp waiwera_amd/$v.so waiwera_amd/libwaiwera_hip.so
  for cfg in c4 c5; do
    (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$v$cfg -o p -- python $GRAFT_REPO_ROOT/bench.py --config $cfg --lead 1 --steps 3 --warmup 0 --no-cpu --spmv-reps 5 > /dev/null 2>&1)
    db=$(find /tmp/prof_$v$cfg -name "*.db" | head -1)
    if [ -n "$db" ]; then python tools/rocprof_summary.py "$db" /tmp/stats_$v$cfg.txt; echo "$v $cfg: $(grep k_dilu_pivots /tmp/stats_$v$cfg.txt | head -1)" >> $out; fi
  done
done
cp waiwera_amd/lib_piv1.so waiwera_amd/libwaiwera_hip.so
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_pc.py tests/test_hip_fullsize.py -m gpu -x -q -k "wce or wae or c4 or c5 or pc" 2>&1 | tail -3 >> $out
cp /tmp/lib_keep.so waiwera_amd/libwaiwera_hip.so
cat $out
