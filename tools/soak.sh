#!/bin/bash
# long runs (TAG=r5 bash tools/soak.sh) of the default path -- the in-launch finalisers (sentinel arrival, bounded waits) and the posted
# scalars over tens of thousands of Krylov iterations back to back; any lost partial sum would end a solve with reason -9
mkdir -p gpurun_out/${TAG:-r5}
export TMPDIR=/tmp
for spec in "c3 300" "c4 300" "c5 1500"; do set -- $spec
  python bench.py --config $1 --steps $2 --warmup 5 --no-cpu 2> gpurun_out/${TAG:-r5}/soak_$1.log | tail -1 > gpurun_out/${TAG:-r5}/soak_$1.json
  python - <<PY
import json, collections
d = json.load(open("gpurun_out/${TAG:-r5}/soak_$1.json")); c = d["config"]
reasons = collections.Counter(r["reason"] for r in c["timed_newton_steps"])
print("soak $1: %d Newton steps, %d Krylov iterations, %.2f steps/s, %.4f ms/it, reasons %s, check %s, device %s"
      % (d["steps"], c["krylov_iterations"], d["value"], c["ms_per_krylov_iteration"], dict(reasons), d["check"].get("passed"),
         d.get("device_state_after_timed_region")))
PY
done 2>&1 | tee gpurun_out/${TAG:-r5}/soak.log
