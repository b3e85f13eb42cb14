# A/B of prebuilt libraries on the same box (kernel microbench only)
# usage: bash tools/ab_libs.sh <config> <reps> <lib name>...      (waiwera_amd/<name>.so)
CFG=$1; REPS=$2; shift 2
cp waiwera_amd/libwaiwera_hip.so /tmp/lib_keep.so
for rep in $(seq $REPS); do
  for v in "$@"; do
    cp waiwera_amd/$v.so waiwera_amd/libwaiwera_hip.so
    echo "$v: $(timeout 300 python bench.py --config $CFG --micro-only 2>&1 | grep -E '^micro')"
  done
done
cp /tmp/lib_keep.so waiwera_amd/libwaiwera_hip.so
