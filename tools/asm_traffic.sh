#!/bin/bash
# Assembly sweeps: duration (rocprofv3 kernel trace) and L2-miss traffic (separate FETCH_SIZE / WRITE_SIZE passes) of
# k_residual / k_residual_tile / k_jacobian_park for one config, with WAI_RES_TILE=0 and 1.
# usage: bash tools/asm_traffic.sh <tag> <config> [bench args]
TAG=$1; CFG=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
ARGS="--config $CFG --lead 1 --steps 2 --warmup 0 --no-cpu --spmv-reps 5 $@"
for t in 0 1; do
  N=${TAG}_${CFG}_tile$t
  rm -rf /tmp/at_$N
  WAI_RES_TILE=$t rocprofv3 --kernel-trace --stats -d /tmp/at_$N/kt -o p -- python bench.py $ARGS > gpurun_out/r4/at_$N.log 2>&1
  python tools/rocprof_summary.py /tmp/at_$N/kt/p_results.db gpurun_out/r4/asm_kernels_$N.txt > /dev/null
  WAI_RES_TILE=$t rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/at_$N/fe -o p -- python bench.py $ARGS >> gpurun_out/r4/at_$N.log 2>&1
  python tools/rocprof_summary.py /tmp/at_$N/fe/p_results.db gpurun_out/r4/asm_fetch_$N.txt --pmc > /dev/null
  WAI_RES_TILE=$t rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/at_$N/wr -o p -- python bench.py $ARGS >> gpurun_out/r4/at_$N.log 2>&1
  python tools/rocprof_summary.py /tmp/at_$N/wr/p_results.db gpurun_out/r4/asm_write_$N.txt --pmc > /dev/null
  echo "== $N"
  grep -E "k_residual|k_jacobian|k_eos" gpurun_out/r4/asm_kernels_$N.txt | cut -c1-120
  python - <<PY
import re
def tab(p):
    o={}
    for ln in open(p):
        f=[x.strip() for x in ln.split("|")]
        if len(f)>=4 and f[1] in ("FETCH_SIZE","WRITE_SIZE"): o[f[0]]=float(f[3])
    return o
fe,wr=tab("gpurun_out/r4/asm_fetch_$N.txt"),tab("gpurun_out/r4/asm_write_$N.txt")
for k in fe:
    if re.search("k_residual|k_jacobian|k_eos", k) and k in wr:
        print("  %-40s 2*FETCH %.3f GB + WRITE %.3f GB = %.3f GB per launch" % (k, 2*fe[k]*1024/1e9, wr[k]*1024/1e9, (2*fe[k]+wr[k])*1024/1e9))
PY
done
