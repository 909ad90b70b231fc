#!/bin/bash
# S-512 eikonal kernel: one / two wavefronts per workgroup, eager / lazy back-pointers on the hybrid heap (same box)
export DAZIM_LIB=/tmp/libdazim_hl.so
DAZIM_HIPCC_EXTRA="-DDZ_FMM_HYBLAZY" python -c "import dazimsurftomo_amd as dz; dz.build(force=True)" > /dev/null 2>&1 || echo build failed
for d in 1 0; do NX=105 OPTS=fmm.dual=$d python tools/fmm_only.py ${1:-1000} 1 2>&1 | grep -E "kernel|checksum" | sed "s/^/lazy dual=$d /"; done
unset DAZIM_LIB
for d in 1 0; do NX=105 OPTS=fmm.dual=$d python tools/fmm_only.py ${1:-1000} 1 2>&1 | grep -E "kernel|checksum" | sed "s/^/eager dual=$d /"; done
