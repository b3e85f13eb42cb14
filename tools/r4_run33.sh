#!/bin/bash
# round 4, pass 33: what the scalar loads ahead of a brick's first vector load cost k_pc_park (list entry -> sub_ptr / sub_nlev):
# a timing-only build that computes them (v_uni, -DWAI_EXP_UNIFORM: valid on a mesh of full bricks only) against the library
# (v_tab), 208^3 = 13 x 13 x 104 full bricks of 16 x 16 x 2, alternating on one box
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
L=gpurun_out/r4/brick_header_ab.log
cp waiwera_amd/libwaiwera_hip.so /tmp/lib_keep.so
for rep in 1 2 3; do for v in v_tab v_uni; do
  cp waiwera_amd/$v.so waiwera_amd/libwaiwera_hip.so
  python bench.py --micro-only --config c3 --dims 208 208 208 --spmv-reps 100 2>&1 | grep '^micro.*\(\[k_pc\|iteration\)' | sed "s/^micro/$v/" | cut -c1-200 | tee -a $L
done; done
cp /tmp/lib_keep.so waiwera_amd/libwaiwera_hip.so
