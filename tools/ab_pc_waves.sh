# A/B of the fused preconditioner kernel's register budget (waves per SIMD) on a real MI355X
for w in 4 5 6; do
  WAI_EXTRA_HIPCC_FLAGS="-DPC_MIN_WAVES=$w" python -m waiwera_amd.build --force > /dev/null 2>&1
  echo "PC_MIN_WAVES=$w"
  python bench.py --dims 160 160 160 --steps 1 --warmup 0 --no-cpu 2>&1 | grep -E "microbench"
done
