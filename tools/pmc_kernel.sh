# PMC counters of one kernel class for one bench configuration (micro-only run), one pass per counter set
# usage: bash tools/pmc_kernel.sh <config> <kernel name substring> "<counters pass 1>" "<counters pass 2>" ...
CFG=$1; KN=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$i -o p -- python bench.py --config $CFG --micro-only --spmv-reps 10 > /tmp/pmc_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  python - "$f" "$KN" <<'PY'
import csv, sys, collections
f, kn = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if kn in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (s, n) in sorted(acc.items()):
    print("%-28s mean %.6g over %d dispatches" % (k, s / n, n))
PY
done
