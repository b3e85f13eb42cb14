#!/bin/bash
# round 6: where the fused launches' traffic above the algorithmic bytes comes from -- L2 / L1 / address-unit counters of
# k_pc_park on a stored operand (first half of an iteration) and with the operand composed in the launch (second half),
# per variant (environment switches / bench args), separate --pmc passes of a micro-only run.
# usage: bash tools/r6_pmc_fused.sh <tag> "<bench args>" "<env A>" "<env B>" ...
TAG=$1; ARGS=$2; shift 2
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_fused_$TAG.txt; : > $OUT
for v in "$@"; do
  i=0
  for set in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
             "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_READ_sum TCC_STREAMING_REQ_sum" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum" \
             "TA_FLAT_READ_WAVEFRONTS_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
             "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rm -rf /tmp/pf_$i
    env $v timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pf_$i -o p -- python bench.py $ARGS --micro-only --spmv-reps 5 > /tmp/pf_$i.log 2>&1
    f=$(find /tmp/pf_$i -name "*counter_collection.csv" | head -1)
    echo "== [$v] pass $i: $set" >> $OUT
    if [ -z "$f" ]; then echo "(no counter file: $(tail -2 /tmp/pf_$i.log | cut -c1-300))" >> $OUT; continue; fi
    python - "$f" >> $OUT <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    kn = r["Kernel_Name"]
    for want in ("k_pc_park<true, true", "k_pc_park<true, false", "k_spmv", "k_bcgs_xrp"):
        if want in kn:
            key = kn[kn.find("k_"):][:44] if "k_pc_park" in kn else want
            a = acc[(key, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (s, n) in sorted(acc.items()):
    print("%-46s %-34s mean %.6g over %d dispatches" % (k, c, s / n, n))
PY
  done
done
cat $OUT
