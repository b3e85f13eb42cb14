#!/bin/bash
# Phase timing of the fused preconditioner kernels: libraries built with -DWAI_PC_PHASES (waiwera_amd/lib_phases*.so,
# see kernels_linalg.hip PH()), kernel microbench only.  usage: bash tools/pc_phases.sh  (writes gpurun_out/pc_phases.log)
# build first, here (the variants travel with the snapshot; they are git-ignored):
#   cd waiwera_amd && WAI_EXTRA_HIPCC_FLAGS="-DWAI_PC_PHASES" python build.py --force && cp libwaiwera_hip.so lib_phases.so
#   WAI_EXTRA_HIPCC_FLAGS="-DWAI_PC_PHASES -DWAI_PC_WAVE=0" python build.py --force && cp libwaiwera_hip.so lib_phases_rows.so
#   python build.py --force
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
cp waiwera_amd/libwaiwera_hip.so /tmp/lib_keep.so
out=gpurun_out/pc_phases.log; : > $out
run() { # lib, label, bench args
  lib=$1; label=$2; shift 2
  cp waiwera_amd/$lib.so waiwera_amd/libwaiwera_hip.so
  echo "== $label ($lib; $*)" >> $out
  timeout 400 python bench.py --micro-only "$@" 2>&1 | grep -E "^micro|pc phases" >> $out
}
run lib_phases c4_rows --config c4
run lib_phases c4_wave_8x4x2 --config c4 --brick 8 4 2
run lib_phases_rows c4_rows_8x4x2 --config c4 --brick 8 4 2
run lib_phases c5_wave --config c5
run lib_phases_rows c5_rows --config c5
cp /tmp/lib_keep.so waiwera_amd/libwaiwera_hip.so
cat $out
