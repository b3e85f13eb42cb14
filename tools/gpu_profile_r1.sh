# Round-1 evidence run on one MI355X: default bench line, rocprofv3 kernel-trace stats of the
# same command, and separate PMC passes (FETCH_SIZE / WRITE_SIZE) for HBM traffic at 216^3.
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_r1.json 2> gpurun_out/bench_r1.log
tail -2 gpurun_out/bench_r1.log | cut -c1-400
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o r1 -- python bench.py --no-cpu > gpurun_out/prof_kt.log 2>&1
python tools/rocprof_summary.py /tmp/prof_kt/r1_results.db gpurun_out/rocprof_kernel_stats_r1.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/prof_fetch -o r1 -- python bench.py --steps 1 --warmup 0 --no-cpu --spmv-reps 10 > gpurun_out/prof_fetch.log 2>&1
python tools/rocprof_summary.py /tmp/prof_fetch/r1_results.db gpurun_out/rocprof_pmc_fetch_r1.txt --pmc
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/prof_write -o r1 -- python bench.py --steps 1 --warmup 0 --no-cpu --spmv-reps 10 > gpurun_out/prof_write.log 2>&1
python tools/rocprof_summary.py /tmp/prof_write/r1_results.db gpurun_out/rocprof_pmc_write_r1.txt --pmc
du -sh /tmp/prof_kt /tmp/prof_fetch /tmp/prof_write
