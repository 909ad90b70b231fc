#!/bin/bash
# eikonal kernel built for 4 wavefronts per SIMD (<= 128 registers) against the default build, S-256 batch, same box
export DAZIM_LIB=/tmp/libdazim_wpe4.so
DAZIM_HIPCC_EXTRA="-DDZ_FMM_WPE=4" python -c "import dazimsurftomo_amd as dz; dz.build(force=True)" > /dev/null 2>&1 || echo build failed
for r in 1 2; do python tools/fmm_only.py 1000 1 2>&1 | grep -E "kernel" | sed "s/^/wpe4 /"; done
WPC=16 python tools/fmm_only.py 1000 1 2>&1 | grep -E "kernel" | sed "s/^/wpe4 /"
unset DAZIM_LIB
for r in 1 2; do python tools/fmm_only.py 1000 1 2>&1 | grep -E "kernel" | sed "s/^/default /"; done
