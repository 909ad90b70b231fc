#!/bin/bash
# per-phase shader-clock totals of the eikonal marching loop (experiment build, see DZ_FMM_PROF in fmm.hip)
touch dazimsurftomo_amd/csrc/fmm.hip
DAZIM_HIPCC_EXTRA="-DDZ_FMM_PROF" python -c "import dazimsurftomo_amd as dz; dz.build()" > /dev/null 2>&1 || echo build failed
timeout 300 python tools/fmm_only.py 1000 1 2>&1 | grep -E "prof|kernel"
touch dazimsurftomo_amd/csrc/fmm.hip
python -c "import dazimsurftomo_amd as dz; dz.build()" > /dev/null 2>&1
