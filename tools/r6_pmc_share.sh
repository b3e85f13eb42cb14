#!/bin/bash
# round 6: the fused launches' wait / in-flight counters at the small sizes (C2 = 100^3, one rank's share of eight = 108^3)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/pmc_fused_small_r6.txt; : > $OUT
for args in "--config c2" "--config c3 --rank-share 8"; do
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
             "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_CYCLE_sum TCC_HIT_sum"; do
    rm -rf /tmp/ps
    timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/ps -o p -- python bench.py $args --micro-only --spmv-reps 3 > /tmp/ps.log 2>&1
    f=$(find /tmp/ps -name "*counter_collection.csv" | head -1)
    echo "== [$args] $set" >> $OUT
    python - "$f" >> $OUT <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    kn = r["Kernel_Name"]
    for want in ("k_pc_park<true, true", "k_pc_park<true, false", "k_spmv", "k_bcgs_xrp"):
        if want in kn:
            a = acc[(want, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (s, n) in sorted(acc.items()):
    print("%-24s %-26s mean %.6g over %d dispatches" % (k, c, s / n, n))
PY
  done
done
cat $OUT
