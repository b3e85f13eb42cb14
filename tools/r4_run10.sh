#!/bin/bash
# round 4, pass 10: (a) what the write-through of the partial sums costs the producers (plain stores, no finaliser);
# (b) base records shared in LDS in k_jacobian_park (WAI_JAC_SHARE_BASE): identity test, duration, traffic at c3
mkdir -p gpurun_out/r4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_parity.py -x -q -k "kernels_agree" 2>&1 | grep -v amdgpu | tail -3 | tee gpurun_out/r4/pytest_10.log
L=gpurun_out/r4/plain_partials.log
cp waiwera_amd/libwaiwera_hip.so /tmp/lib_keep.so
for rep in 1 2; do for v in v_default v_plain; do
  cp waiwera_amd/$v.so waiwera_amd/libwaiwera_hip.so
  for cfg in c4 c5 c3; do
    echo "$v $(WAI_MICRO_NOFIN=1 python bench.py --micro-only --config $cfg --spmv-reps 100 2>&1 | grep '^micro')" | tee -a $L
  done
done; done
cp /tmp/lib_keep.so waiwera_amd/libwaiwera_hip.so
ARGS="--config c3 --lead 1 --steps 2 --warmup 0 --no-cpu --spmv-reps 5"
for sb in 0 1; do
  N=sb$sb
  rm -rf /tmp/jb_$N
  WAI_JAC_SHARE_BASE=$sb rocprofv3 --kernel-trace --stats -d /tmp/jb_$N/kt -o p -- python bench.py $ARGS > gpurun_out/r4/jb_$N.log 2>&1
  python tools/rocprof_summary.py /tmp/jb_$N/kt/p_results.db gpurun_out/r4/jb_kernels_$N.txt > /dev/null
  WAI_JAC_SHARE_BASE=$sb rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/jb_$N/fe -o p -- python bench.py $ARGS >> gpurun_out/r4/jb_$N.log 2>&1
  python tools/rocprof_summary.py /tmp/jb_$N/fe/p_results.db gpurun_out/r4/jb_fetch_$N.txt --pmc > /dev/null
  WAI_JAC_SHARE_BASE=$sb rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/jb_$N/wr -o p -- python bench.py $ARGS >> gpurun_out/r4/jb_$N.log 2>&1
  python tools/rocprof_summary.py /tmp/jb_$N/wr/p_results.db gpurun_out/r4/jb_write_$N.txt --pmc > /dev/null
  echo "== WAI_JAC_SHARE_BASE=$sb"
  grep -E "k_jacobian" gpurun_out/r4/jb_kernels_$N.txt | cut -c1-120
  grep -E "k_jacobian" gpurun_out/r4/jb_fetch_$N.txt gpurun_out/r4/jb_write_$N.txt | cut -c1-160
done
