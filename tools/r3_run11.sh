#!/bin/bash
micro() { python bench.py "$@" --micro-only 2>&1 | grep "^micro" | cut -c1-200; }
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -Wno-unused-result tools/micro/spmv3_variants.hip -o /tmp/spmv3_variants 2>/dev/null; /tmp/spmv3_variants
echo "== library, default"; micro --config c4; micro --config c4 --cell-order natural; micro
WAI_EXTRA_HIPCC_FLAGS="-DWAI_SPMV_NOREMAP" python -m waiwera_amd.build --force > /dev/null 2>&1
echo "== library, k_spmv without the XCD remap"; micro --config c4; micro --config c4 --cell-order natural; micro
python -m waiwera_amd.build --force > /dev/null 2>&1
