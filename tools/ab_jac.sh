# k_jacobian against k_jacobian_park (WAI_JAC_PARK=0 / 1): kernel time from a rocprofv3 kernel trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in "$@"; do
  for p in 0 1; do
    rm -rf /tmp/jp_$p
    WAI_JAC_PARK=$p rocprofv3 --kernel-trace --stats -d /tmp/jp_$p -o p -- python bench.py --config $c --no-cpu --lead 1 --steps 3 --warmup 0 > /tmp/jp_$p.log 2>&1
    python tools/rocprof_summary.py /tmp/jp_$p/p_results.db /tmp/jp_$p.txt > /dev/null 2>&1
    echo "$c PARK=$p: $(grep -E 'k_jacobian' /tmp/jp_$p.txt | cut -c1-100)"
  done
done
