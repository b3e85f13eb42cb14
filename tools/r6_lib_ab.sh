#!/bin/bash
# same-box A/B of a prebuilt library variant against the base: bits (pc / parity tests on the variant), then C3, the 108^3 share, C2
# usage: tools/r6_lib_ab.sh <name>      (waiwera_amd/lib_<name>.so and lib_base.so, built with WAI_EXTRA_HIPCC_FLAGS)
V=$1
cp waiwera_amd/libwaiwera_hip.so /tmp/keep.so
cp waiwera_amd/lib_$V.so waiwera_amd/libwaiwera_hip.so
python -m pytest tests/test_hip_pc.py tests/test_hip_parity.py -q -m gpu -p no:cacheprovider 2>&1 | tail -2
cp /tmp/keep.so waiwera_amd/libwaiwera_hip.so
bash tools/ab.sh ${V}_ab_r6_c3 3 "--config c3 --micro-only --spmv-reps 100" "LIB=lib_base" "LIB=lib_$V" | grep -E "==|as an iteration|device-only"
bash tools/ab.sh ${V}_ab_r6_share8 3 "--config c3 --rank-share 8 --micro-only --spmv-reps 200" "LIB=lib_base" "LIB=lib_$V" | grep -E "==|as an iteration|device-only"
bash tools/ab.sh ${V}_ab_r6_c2 2 "--config c2 --micro-only --spmv-reps 200" "LIB=lib_base" "LIB=lib_$V" | grep -E "==|as an iteration|device-only"
