#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3
cp waiwera_amd/libwaiwera_hip.so /tmp/lib_keep.so
out=gpurun_out/r3/run18.log; : > $out
for rep in 1 2; do
for v in "$@"; do
  cp waiwera_amd/$v.so waiwera_amd/libwaiwera_hip.so
  echo "== $v" >> $out
  timeout 400 python bench.py --micro-only --config ${CFG:-c4} 2>&1 | grep -E "^micro|pc phases" >> $out
done
done
cp /tmp/lib_keep.so waiwera_amd/libwaiwera_hip.so
cat $out
