cp waiwera_amd/libwaiwera_hip.so /tmp/keep.so
cp waiwera_amd/lib_epiw.so waiwera_amd/libwaiwera_hip.so
python -m pytest tests/test_hip_pc.py tests/test_hip_parity.py tests/test_hip_tracer.py -q -m gpu -p no:cacheprovider 2>&1 | tail -2
cp /tmp/keep.so waiwera_amd/libwaiwera_hip.so
bash tools/ab.sh epiw_ab_r6_c3 3 "--config c3 --micro-only --spmv-reps 100" "LIB=lib_base" "LIB=lib_epiw" | grep -E "==|as an iteration|device-only"
bash tools/ab.sh epiw_ab_r6_share8 3 "--config c3 --rank-share 8 --micro-only --spmv-reps 200" "LIB=lib_base" "LIB=lib_epiw" | grep -E "==|as an iteration|device-only"
bash tools/ab.sh epiw_ab_r6_c4 2 "--config c4 --micro-only --spmv-reps 100" "LIB=lib_base" "LIB=lib_epiw" | grep -E "==|as an iteration|device-only"
