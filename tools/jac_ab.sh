# duration of the Jacobian sweep, row-wise (k_jacobian_park) against column-wise (k_jacobian_sym): rocprofv3 kernel stats of a short bench run each
# usage: tools/jac_ab.sh <tag> <config>...
TAG=$1; shift
export TMPDIR=/tmp
for cfg in "$@"; do for v in 0 1; do
  d=/tmp/jacab_${cfg}_$v; rm -rf $d
  (cd /tmp && WAI_JAC_SYM=$v rocprofv3 --kernel-trace --stats -d $d -o p -- python $GRAFT_REPO_ROOT/bench.py --config $cfg --lead 1 --steps 4 --warmup 1 --no-cpu --spmv-reps 3 > /dev/null 2>&1)
  python tools/rocprof_summary.py $d/p_results.db /tmp/jacab_$cfg_$v.txt > /dev/null
  echo "== $cfg WAI_JAC_SYM=$v"; grep -E "k_jacobian|k_residual_tile|k_eos_pert" /tmp/jacab_$cfg_$v.txt | cut -c1-140
done; done | tee gpurun_out/jac_ab_$TAG.log
