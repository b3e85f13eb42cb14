#!/bin/bash
# round 6: k_pc_park as a persistent launch (768 workgroups that walk their XCD's bricks) against one workgroup per brick
mkdir -p gpurun_out
cp waiwera_amd/libwaiwera_hip.so /tmp/keep.so
cp waiwera_amd/lib_loop.so waiwera_amd/libwaiwera_hip.so
WAI_PC_PERSIST=1 python -m pytest tests/test_hip_pc.py tests/test_hip_parity.py -q -m gpu -p no:cacheprovider 2>&1 | tail -2
cp /tmp/keep.so waiwera_amd/libwaiwera_hip.so
bash tools/ab.sh persist_ab_r6_c3 3 "--config c3 --micro-only --spmv-reps 100" "LIB=lib_base" "LIB=lib_loop WAI_PC_PERSIST=0" "LIB=lib_loop WAI_PC_PERSIST=1" | grep -E "==|as an iteration|device-only"
bash tools/ab.sh persist_ab_r6_share8 3 "--config c3 --rank-share 8 --micro-only --spmv-reps 200" "LIB=lib_base" "LIB=lib_loop WAI_PC_PERSIST=0" "LIB=lib_loop WAI_PC_PERSIST=1" | grep -E "==|as an iteration|device-only"
bash tools/ab.sh persist_ab_r6_c2 2 "--config c2 --micro-only --spmv-reps 200" "LIB=lib_base" "LIB=lib_loop WAI_PC_PERSIST=1" | grep -E "==|as an iteration|device-only"
