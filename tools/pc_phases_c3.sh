#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
cp waiwera_amd/libwaiwera_hip.so /tmp/lib_keep.so
out=gpurun_out/pc_phases_c3.log; : > $out
cp waiwera_amd/lib_phases.so waiwera_amd/libwaiwera_hip.so
timeout 600 python bench.py --micro-only --config c3 2>&1 | grep -E "^micro|pc phases" >> $out
timeout 600 python bench.py --micro-only --config c3 --rank-share 8 2>&1 | grep -E "^micro|pc phases" >> $out
cp /tmp/lib_keep.so waiwera_amd/libwaiwera_hip.so
cat $out
