# fused-kernel time against brick shape, for prebuilt libraries (waiwera_amd/<name>.so)
# usage: bash tools/ab_bricks.sh <config> "<bx by bz>;<bx by bz>;..." <lib name>...
CFG=$1; BR=$2; shift 2
cp waiwera_amd/libwaiwera_hip.so /tmp/lib_keep.so
IFS=';' read -ra BRS <<< "$BR"
for b in "${BRS[@]}"; do
  for v in "$@"; do
    cp waiwera_amd/$v.so waiwera_amd/libwaiwera_hip.so
    echo "$v brick $b: $(timeout 300 python bench.py --config $CFG --micro-only --brick $b 2>&1 | grep -E '^micro' | cut -d: -f2-)"
  done
done
cp /tmp/lib_keep.so waiwera_amd/libwaiwera_hip.so
