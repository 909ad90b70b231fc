#!/bin/bash
# hit rate of the two-wavefront eikonal kernel's speculative requests (experiment build, DZ_FMM_DUALSTAT in fmm.hip)
export DAZIM_LIB=/tmp/libdazim_dualstat.so
DAZIM_HIPCC_EXTRA="-DDZ_FMM_DUALSTAT" python -c "import dazimsurftomo_amd as dz; dz.build(force=True)" > /dev/null 2>&1 || echo build failed
timeout 300 python tools/fmm_only.py ${1:-200} 1 2>&1 | grep -E "dual|kernel"
