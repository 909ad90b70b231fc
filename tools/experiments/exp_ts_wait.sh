#!/bin/bash
# how long the workgroups of the time-sliced eikonal kernel wait for the previous stage of the task they took (DZ_TS_WAITSTAT build)
export DAZIM_LIB=/tmp/libdazim_tsw.so
DAZIM_HIPCC_EXTRA="-DDZ_TS_WAITSTAT" python -c "import dazimsurftomo_amd as dz; dz.build(force=True)" > /dev/null 2>&1 || echo build failed
for o in fmm.ts_stages=2 fmm.ts_stages=4 fmm.ts_stages=8; do
  OPTS=$o python tools/fmm_only.py ${1:-1000} 1 2>&1 | grep -E "kernel|ts wait" | tr "\n" " " | sed "s/^/$o /"; echo
done
