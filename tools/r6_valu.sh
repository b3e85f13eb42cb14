#!/bin/bash
# round 6 ("report FP64 VALU utilisation next to GB/s" for the EOS-heavy sweeps): SQ counters of the assembly
# kernels in a micro-only run of one config, separate --pmc passes, kernel-trace only.
# usage: bash tools/r6_valu.sh <tag> <config>
TAG=$1; CFG=$2
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/valu_${TAG}_${CFG}.txt; : > $OUT
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]+|GRBM_[A-Z_0-9]+|VALUBusy|VALUUtilization|OccupancyPercent|MemUnitBusy|MemUnitStalled)\b" | sort -u > gpurun_out/counters_available_$TAG.txt
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
           "VALUBusy VALUUtilization" "MemUnitBusy MemUnitStalled"; do
  i=$((i+1))
  rm -rf /tmp/valu_$i
  timeout 900 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/valu_$i -o p -- python bench.py --config $CFG --micro-only --spmv-reps 5 > /tmp/valu_$i.log 2>&1
  f=$(find /tmp/valu_$i -name "*counter_collection.csv" | head -1)
  echo "== pass $i: $set" >> $OUT
  if [ -z "$f" ]; then echo "(no counter file: $(tail -2 /tmp/valu_$i.log | cut -c1-300))" >> $OUT; continue; fi
  python - "$f" >> $OUT <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    kn = r["Kernel_Name"]
    for want in ("k_eos_pert", "k_eos<", "k_jacobian_sym", "k_residual_tile", "k_pc_park<true, true", "k_pc_park<true, false", "k_spmv"):
        if want in kn:
            a = acc[(want, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (s, n) in sorted(acc.items()):
    print("%-24s %-24s mean %.6g over %d dispatches" % (k, c, s / n, n))
PY
done
cat $OUT
