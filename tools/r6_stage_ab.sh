#!/bin/bash
# round 6: k_pc_park's operand stage (WAI_PC_STAGE=0 | 1 | 2) -- bits (the preconditioner tests under each setting), then
# the same-box A/B at C3, the 108^3 rank share and C2
mkdir -p gpurun_out
for st in 1 2; do
  WAI_PC_STAGE=$st timeout 900 python -m pytest tests/test_hip_pc.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/stage_tests_$st.log
  cat gpurun_out/stage_tests_$st.log
done
bash tools/ab.sh stage_ab_r6_c3 3 "--config c3 --micro-only --spmv-reps 100" "WAI_PC_STAGE=0" "WAI_PC_STAGE=1" "WAI_PC_STAGE=2"
bash tools/ab.sh stage_ab_r6_share8 3 "--config c3 --rank-share 8 --micro-only --spmv-reps 200" "WAI_PC_STAGE=0" "WAI_PC_STAGE=1" "WAI_PC_STAGE=2"
bash tools/ab.sh stage_ab_r6_c2 2 "--config c2 --micro-only --spmv-reps 200" "WAI_PC_STAGE=0" "WAI_PC_STAGE=1" "WAI_PC_STAGE=2"
