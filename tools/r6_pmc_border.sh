#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/pmc_fused_border_r6_c3.txt; : > $OUT
for bo in x tile4x4 tile2x2; do
  rm -rf /tmp/pb
  timeout 900 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d /tmp/pb -o p -- python bench.py --config c3 --micro-only --spmv-reps 3 --brick-order $bo > /tmp/pb.log 2>&1
  f=$(find /tmp/pb -name "*counter_collection.csv" | head -1)
  echo "== --brick-order $bo" >> $OUT
  python - "$f" >> $OUT <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    kn = r["Kernel_Name"]
    for want in ("k_pc_park<true, true", "k_pc_park<true, false", "k_spmv", "k_jacobian_sym", "k_residual_tile"):
        if want in kn:
            a = acc[(want, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (s, n) in sorted(acc.items()):
    print("%-24s %-22s mean %.6g over %d dispatches" % (k, c, s / n, n))
PY
done
cat $OUT
