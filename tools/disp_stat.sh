#!/bin/bash
# how much of the lane-evaluations the dispersion kernel's wavefronts / workgroups offer is used (experiment build, DZ_DISP_STAT)
export DAZIM_LIB=/tmp/libdazim_dstat.so
DAZIM_HIPCC_EXTRA="-DDZ_DISP_STAT" python -c "import dazimsurftomo_amd as dz; dz.build(force=True)" > /dev/null 2>&1 || echo build failed
python tools/disp_only.py ${1:-54} 2>&1 | grep -E "disp stat|kernel" | tail -3
