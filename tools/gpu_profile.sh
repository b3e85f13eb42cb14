# Evidence run on one MI355X for one bench configuration: the default bench line, rocprofv3
# kernel-trace stats of the same command, and separate PMC passes (FETCH_SIZE / WRITE_SIZE) for HBM
# traffic.  usage: bash tools/gpu_profile.sh <round tag, e.g. r2> <config c2..c5> [extra bench args]
set -x
TAG=$1; CFG=$2; shift 2
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
N=${TAG}_${CFG}
python bench.py --config $CFG "$@" > gpurun_out/bench_$N.json 2> gpurun_out/bench_$N.log
tail -2 gpurun_out/bench_$N.log | cut -c1-400
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt_$N -o p -- python bench.py --config $CFG --no-cpu "$@" > gpurun_out/prof_kt_$N.log 2>&1
python tools/rocprof_summary.py /tmp/prof_kt_$N/p_results.db gpurun_out/rocprof_kernel_stats_$N.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/prof_fetch_$N -o p -- python bench.py --config $CFG --lead 1 --steps 1 --warmup 0 --no-cpu --spmv-reps 10 "$@" > gpurun_out/prof_fetch_$N.log 2>&1
python tools/rocprof_summary.py /tmp/prof_fetch_$N/p_results.db gpurun_out/rocprof_pmc_fetch_$N.txt --pmc
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/prof_write_$N -o p -- python bench.py --config $CFG --lead 1 --steps 1 --warmup 0 --no-cpu --spmv-reps 10 "$@" > gpurun_out/prof_write_$N.log 2>&1
python tools/rocprof_summary.py /tmp/prof_write_$N/p_results.db gpurun_out/rocprof_pmc_write_$N.txt --pmc
python tools/pmc_traffic.py gpurun_out $N gpurun_out/pmc_traffic_$N.json
