# C3 at full size and its eight-rank share on ONE box (one gpurun call): the ratio share / (full / 8) is box-dependent
# (the boxes differ at full size, hardly at the share).   TAG=r5k1 bash tools/share_pair.sh
TAG=${TAG:-r5k}
python bench.py --steps 12 --warmup 3 --no-cpu --spmv-reps 50 2>/dev/null > gpurun_out/bench_${TAG}_c3.json
python bench.py --rank-share 8 --steps 12 --warmup 3 --no-cpu --spmv-reps 50 2>/dev/null > gpurun_out/bench_${TAG}_c3_share8.json
python - <<PY
import json
a=json.load(open("gpurun_out/bench_${TAG}_c3.json")); b=json.load(open("gpurun_out/bench_${TAG}_c3_share8.json"))
f=a["config"]["ms_per_krylov_iteration_device_only"]; s=b["config"]["ms_per_krylov_iteration_device_only"]
print("${TAG}: C3 %.4f ms per iteration (fused %.1f %%), share8 %.4f ms (fused %.1f %%): %.3f x the eighth" % (f, 100*a["roofline"]["frac"], s, 100*b["roofline"]["frac"], s/(f/8)))
PY
