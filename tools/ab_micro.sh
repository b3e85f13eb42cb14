#!/bin/bash
# same-box A/B of prebuilt library variants on the kernel microbench of several configs
# usage: bash tools/ab_micro.sh "<cfg> ..." <reps> <lib name>...      (waiwera_amd/<name>.so; writes gpurun_out/ab_micro.log)
cd "$(dirname "$0")/.."
CFGS=$1; REPS=$2; shift 2
mkdir -p gpurun_out
cp waiwera_amd/libwaiwera_hip.so /tmp/lib_keep.so
out=gpurun_out/ab_micro.log; : > $out
for rep in $(seq $REPS); do
  for cfg in $CFGS; do
    for v in "$@"; do
      cp waiwera_amd/$v.so waiwera_amd/libwaiwera_hip.so
      echo "$v: $(timeout 400 python bench.py --micro-only --config $cfg 2>&1 | grep -E '^micro')" >> $out
    done
  done
done
cp /tmp/lib_keep.so waiwera_amd/libwaiwera_hip.so
cat $out
