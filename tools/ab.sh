#!/bin/bash
# Same-box A/B of kernel variants selected by environment switches or by prebuilt libraries (ONE gpurun call):
#   tools/ab.sh <log name> <rounds> "<bench args>" "<env of variant A>" "<env of variant B>" ...
# e.g. tools/ab.sh compose_r5 3 "--rank-share 8 --micro-only --spmv-reps 200" "WAI_BCGS_COMPOSE=0" "WAI_BCGS_COMPOSE=1"
#      tools/ab.sh idx16_r5 3 "--config c3 --micro-only" "LIB=lib_base" "LIB=lib_idx16"      (waiwera_amd/<name>.so, prebuilt)
# The variants run alternately <rounds> times (boxes drift with temperature: alternate, do not batch); the lines
# bench.py --micro-only logs (kernel times, the iteration's device-only time) go to gpurun_out/<log name>.log.
NAME=$1; ROUNDS=$2; ARGS=$3; shift 3
mkdir -p gpurun_out
: > gpurun_out/$NAME.log
cp waiwera_amd/libwaiwera_hip.so /tmp/lib_keep.so
for r in $(seq $ROUNDS); do
  for v in "$@"; do
    echo "== round $r: $v" >> gpurun_out/$NAME.log
    for tok in $v; do case $tok in LIB=*) cp waiwera_amd/${tok#LIB=}.so waiwera_amd/libwaiwera_hip.so;; esac; done
    env $v python bench.py $ARGS 2>&1 | grep -E "^\{|micro|ms per|device-only" | cut -c1-1200 >> gpurun_out/$NAME.log
  done
done
cp /tmp/lib_keep.so waiwera_amd/libwaiwera_hip.so
cat gpurun_out/$NAME.log
