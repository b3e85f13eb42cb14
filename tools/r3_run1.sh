#!/bin/bash
# round-3 GPU pass: whole GPU suite, then the bench lines (c3, rank shares)
mkdir -p gpurun_out/r3
python -m pytest tests -m gpu -x -q -s --durations=15 > gpurun_out/r3/pytest_gpu.log 2>&1
echo "pytest rc $?" ; tail -5 gpurun_out/r3/pytest_gpu.log
python bench.py --dims 48 48 48 --steps 4 --warmup 1 --lead 1 --window 2 > gpurun_out/r3/bench_small.json 2> gpurun_out/r3/bench_small.log
echo "bench small rc $?"; tail -3 gpurun_out/r3/bench_small.log; cat gpurun_out/r3/bench_small.json | head -c 1500
python bench.py --steps 20 --warmup 5 > gpurun_out/r3/bench_c3.json 2> gpurun_out/r3/bench_c3.log
echo "bench c3 rc $?"; grep -v "^  " gpurun_out/r3/bench_c3.log | tail -12
python bench.py --rank-share 8 --steps 20 --warmup 5 --no-cpu > gpurun_out/r3/bench_c3_share8.json 2> gpurun_out/r3/bench_c3_share8.log
echo "bench c3/8 rc $?"; grep -v "^  " gpurun_out/r3/bench_c3_share8.log | tail -6
python bench.py --config c4 --rank-share 4 --steps 20 --warmup 5 --no-cpu > gpurun_out/r3/bench_c4_share4.json 2> gpurun_out/r3/bench_c4_share4.log
echo "bench c4/4 rc $?"; grep -v "^  " gpurun_out/r3/bench_c4_share4.log | tail -6
