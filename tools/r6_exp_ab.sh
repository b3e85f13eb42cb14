#!/bin/bash
# round 6: two latency probes of k_pc_park, built as separate libraries (WAI_EXTRA_HIPCC_FLAGS): -DWAI_PC_SETPRIO (the sweeps at
# raised wave priority) and -DWAI_PC_EPI2 (no barrier in front of the reduction scratch, one lane per slot for the wave sums' total)
mkdir -p gpurun_out
cp waiwera_amd/libwaiwera_hip.so /tmp/keep.so
cp waiwera_amd/lib_both.so waiwera_amd/libwaiwera_hip.so
python -m pytest tests/test_hip_pc.py tests/test_hip_parity.py -q -m gpu -p no:cacheprovider 2>&1 | tail -2
cp /tmp/keep.so waiwera_amd/libwaiwera_hip.so
bash tools/ab.sh exp_ab_r6_c3 3 "--config c3 --micro-only --spmv-reps 100" "LIB=lib_base" "LIB=lib_prio" "LIB=lib_epi" "LIB=lib_both" | grep -E "==|as an iteration|device-only"
bash tools/ab.sh exp_ab_r6_share8 3 "--config c3 --rank-share 8 --micro-only --spmv-reps 200" "LIB=lib_base" "LIB=lib_prio" "LIB=lib_epi" "LIB=lib_both" | grep -E "==|as an iteration|device-only"
