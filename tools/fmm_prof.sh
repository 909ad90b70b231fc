#!/bin/bash
# per-phase shader-clock totals of the eikonal marching loop (experiment build into its own library, see DZ_FMM_PROF in fmm.hip)
export DAZIM_LIB=/tmp/libdazim_prof.so
DAZIM_HIPCC_EXTRA="-DDZ_FMM_PROF ${PROF_EXTRA:-}" python -c "import dazimsurftomo_amd as dz; dz.build(force=True)" > /dev/null 2>&1 || echo build failed
WPC=${WPC:-0} timeout 300 python tools/fmm_only.py ${1:-1000} 1 2>&1 | grep -E "prof|kernel"
