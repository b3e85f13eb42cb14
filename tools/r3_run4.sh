#!/bin/bash
# round-3 GPU pass 4: per-kernel A/B of the brick order, k_pc_wave against k_pc_rows on the same box, ILU(1) at C3
mkdir -p gpurun_out/r3
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest "tests/test_hip_multirank.py::test_asm_overlap_reaches_across_ranks" tests/test_hip_multirank.py::test_overlapped_halo_exchange_eight_ranks -m gpu -x -q -s --durations=4 2>&1 | tail -8 | cut -c1-200
prof() { name=$1; shift
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o p -- python $R/bench.py "$@" --no-cpu > $R/gpurun_out/r3/prof_$name.json 2> $R/gpurun_out/r3/prof_$name.log)
  db=$(find /tmp/prof_$name -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/rocprof_summary.py "$db" gpurun_out/r3/kstats_$name.txt; head -14 gpurun_out/r3/kstats_$name.txt | cut -c1-150; else echo "no db for $name"; fi
}
prof c3_z --steps 8 --warmup 2 --lead 1
prof c3_x --steps 8 --warmup 2 --lead 1 --brick-order x
prof c4 --config c4 --steps 8 --warmup 2 --lead 1
micro() { python bench.py "$@" --micro-only 2>&1 | grep "^micro" | cut -c1-200; }
echo "== k_pc_wave (default build)"; micro --config c4 --brick 8 4 2; micro --config c5; micro --config c4
WAI_EXTRA_HIPCC_FLAGS="-DWAI_PC_WAVE=0" python -m waiwera_amd.build --force > /dev/null 2>&1
echo "== k_pc_rows (-DWAI_PC_WAVE=0)"; micro --config c4 --brick 8 4 2; micro --config c5; micro --config c4
python -m waiwera_amd.build --force > /dev/null 2>&1
python bench.py --steps 10 --warmup 2 --ilu-levels 1 --no-cpu > gpurun_out/r3/ilu1_c3.json 2> gpurun_out/r3/ilu1_c3.log; python - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/r3/ilu1_c3.json")); c=d["config"]
    print("ilu1_c3 value %.3f acc %s its/step %.1f ms/it %.3f" % (d["value"], d.get("value_accepted_steps"), c["krylov_iterations_per_newton_step"], c["ms_per_krylov_iteration"]))
    print([(r["time_step"], r["dt"], r["krylov"], r["reason"]) for r in c["timed_newton_steps"]])
except Exception as e: print("ilu1_c3:", e)
PY
