#!/bin/bash
# k_pc_rows3: correctness on the wce cases, then phases and A/B on c4
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_pc.py -m gpu -x -q -k "wce or wae or spmv_ilu or pc" 2>&1 | tail -6
cp waiwera_amd/libwaiwera_hip.so /tmp/lib_keep.so
out=gpurun_out/r3/run17.log; : > $out
for rep in 1 2; do
for v in lib_phases lib_rows3 lib_norows3; do
  cp waiwera_amd/$v.so waiwera_amd/libwaiwera_hip.so
  echo "== $v" >> $out
  timeout 400 python bench.py --micro-only --config c4 2>&1 | grep -E "^micro|pc phases" >> $out
done
done
cp /tmp/lib_keep.so waiwera_amd/libwaiwera_hip.so
cat $out
