#!/bin/bash
micro() { python bench.py "$@" --micro-only 2>&1 | grep "^micro" | cut -c1-200; }
python -m pytest tests/test_hip_parity.py tests/test_hip_pc.py tests/test_hip_fullsize.py "tests/test_hip_input.py::test_rock_table_controls" -m gpu -x -q 2>&1 | tail -3
echo "== default build (two-wave bricks for c4)"; micro --config c4; micro --config c4 --cell-order hyperplane; micro --config c4 --rank-share 4
WAI_EXTRA_HIPCC_FLAGS="-DWAI_PC_WAVE=0" python -m waiwera_amd.build --force > /dev/null 2>&1
echo "== -DWAI_PC_WAVE=0 (k_pc_rows)"; micro --config c4; micro --config c4 --cell-order hyperplane; micro --config c4 --rank-share 4
python -m waiwera_amd.build --force > /dev/null 2>&1
