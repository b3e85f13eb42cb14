#!/bin/bash
# round 6: numbering of the bricks (memory and launch order) -- x fastest against columns of A x B bricks ("tileAxB": the
# bricks above / below A * B positions away) and strips of N brick rows ("tileN")
# usage: tools/r6_border_ab.sh <log tag> <config args> <orders...>
mkdir -p gpurun_out
TAG=$1; ARGS=$2; shift 2
L=gpurun_out/border_ab_r6_$TAG.log; : > $L
for r in 1 2; do
  for bo in "$@"; do
    echo "== round $r: --brick-order $bo" >> $L
    python bench.py $ARGS --micro-only --spmv-reps 100 --brick-order $bo 2>&1 | grep -E "micro|ms per|device-only" | cut -c1-600 >> $L
  done
done
grep -E "==|assembly|as an iteration|device-only|spmv 0" $L | cut -c1-230
