set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 6 --warmup 2 > gpurun_out/bench_r1.json 2> gpurun_out/bench_r1.log
tail -3 gpurun_out/bench_r1.log
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt -o r1 -- python bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/prof_kt.log 2>&1
ls -R gpurun_out/prof_kt | head -20
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/prof_fetch -o r1 -- python bench.py --dims 160 160 160 --steps 1 --warmup 0 --no-cpu --spmv-reps 5 > gpurun_out/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/prof_write -o r1 -- python bench.py --dims 160 160 160 --steps 1 --warmup 0 --no-cpu --spmv-reps 5 > gpurun_out/prof_write.log 2>&1
ls -R gpurun_out/prof_fetch | head; du -sh gpurun_out/*
