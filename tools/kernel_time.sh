# average duration of the kernels matching a pattern in a short bench run (rocprofv3 kernel trace)
# usage: bash tools/kernel_time.sh <pattern> <config>...
PAT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in "$@"; do
  rm -rf /tmp/kt_$c
  rocprofv3 --kernel-trace --stats -d /tmp/kt_$c -o p -- python bench.py --config $c --no-cpu --lead 1 --steps 3 --warmup 0 > /tmp/kt_$c.log 2>&1
  python tools/rocprof_summary.py /tmp/kt_$c/p_results.db /tmp/kt_$c.txt > /dev/null 2>&1
  grep -E "$PAT" /tmp/kt_$c.txt | cut -c1-110 | sed "s/^/$c: /"
done
