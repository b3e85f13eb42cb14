#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/pmc_fused_latency_r6_c3.txt; : > $OUT
i=0
for set in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_MULTI_MISS_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_BUSY_sum TCC_CYCLE_sum"; do
  i=$((i+1)); rm -rf /tmp/pl_$i
  timeout 900 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pl_$i -o p -- python bench.py --config c3 --micro-only --spmv-reps 3 > /tmp/pl_$i.log 2>&1
  f=$(find /tmp/pl_$i -name "*counter_collection.csv" | head -1)
  echo "== pass $i: $set" >> $OUT
  if [ -z "$f" ]; then echo "(no counter file: $(tail -2 /tmp/pl_$i.log | cut -c1-300))" >> $OUT; continue; fi
  python - "$f" >> $OUT <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    kn = r["Kernel_Name"]
    for want in ("k_pc_park<true, true", "k_pc_park<true, false", "k_spmv", "k_bcgs_xrp"):
        if want in kn:
            a = acc[(want, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (s, n) in sorted(acc.items()):
    print("%-24s %-34s mean %.6g over %d dispatches" % (k, c, s / n, n))
PY
done
cat $OUT
