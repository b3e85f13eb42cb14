# loopback timing of the multi-rank tests with / without the overlapped halo exchange
for v in 1 0; do
  if [ $v = 1 ]; then export WAI_NO_HALO_OVERLAP=1; else unset WAI_NO_HALO_OVERLAP; fi
  echo "no_overlap=$v"
  for k in "match_one_rank and 2" "match_one_rank and 8" "driver_launches and 2x1x1" "driver_launches and 2x2x2"; do
    s=$(date +%s.%N)
    r=$(timeout 900 python -m pytest tests/test_hip_multirank.py -m gpu -q -k "$k" 2>&1 | grep -E "passed|failed" | tail -1)
    e=$(date +%s.%N)
    echo "  [$k] $r  wall $(python -c "print('%.1f' % ($e - $s))") s"
  done
done
