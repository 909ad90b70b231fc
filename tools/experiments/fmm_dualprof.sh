#!/bin/bash
# stage clocks of the two-wavefront eikonal kernel (experiment build, DZ_FMM_DUALPROF in fmm.hip): tools/fmm_dualprof.sh [sources]
export DAZIM_LIB=/tmp/libdazim_dualprof.so
DAZIM_HIPCC_EXTRA="-DDZ_FMM_DUALPROF" python -c "import dazimsurftomo_amd as dz; dz.build(force=True)" > /dev/null 2>&1 || echo build failed
timeout 300 python tools/fmm_only.py ${1:-200} 1 2>&1 | grep -E "dual|kernel"
