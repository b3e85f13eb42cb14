#!/bin/bash
# The kernels the default build never selects on hexahedral / MINC meshes -- stored-factor ILU(0) (k_ilu_factor, k_pc
# reading the factor), no one-thread-per-scalar-row / one-wave-per-brick kernels, BiCGStab without the speculative half
# iteration, shuffle-tree wave sums -- through the pc / parity / tracer tests.
#   here:        bash tools/ci_fallback_kernels.sh build      (waiwera_amd/v_fallback.so, cross-compiled)
#   on the GPU:  bash tools/ci_fallback_kernels.sh            (swaps the library in, runs the tests, swaps it back)
cd "$(dirname "$0")/.."
FLAGS="-DWAI_ILU_GENERAL -DWAI_PC_ROWS=0 -DWAI_PC_WAVE=0 -DWAI_BCGS_NO_SPECULATION -DWAI_SHFL_SUMS"
if [ "$1" = build ]; then
  cp waiwera_amd/libwaiwera_hip.so /tmp/lib_default.so
  WAI_EXTRA_HIPCC_FLAGS="$FLAGS" python -m waiwera_amd.build --force > /dev/null || exit 1
  cp waiwera_amd/libwaiwera_hip.so waiwera_amd/v_fallback.so
  python -m waiwera_amd.build --force > /dev/null || exit 1
  cmp -s waiwera_amd/libwaiwera_hip.so /tmp/lib_default.so || echo "note: the default library was rebuilt from changed sources"
  exit 0
fi
[ -f waiwera_amd/v_fallback.so ] || { echo "build first"; exit 1; }
cp waiwera_amd/libwaiwera_hip.so /tmp/lib_keep.so
cp waiwera_amd/v_fallback.so waiwera_amd/libwaiwera_hip.so
python -m pytest tests/test_hip_pc.py tests/test_hip_parity.py tests/test_hip_tracer.py tests/test_hip_salt.py -q -x -k "not four_launches and not column_indices" 2>&1 | grep -v amdgpu | tail -6
rc=${PIPESTATUS[0]}
cp /tmp/lib_keep.so waiwera_amd/libwaiwera_hip.so
exit $rc
