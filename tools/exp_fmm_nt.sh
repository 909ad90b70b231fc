#!/bin/bash
# eikonal kernel with non-temporal hints on the node-record loads (1), stores (2), both (3); each build into its own library
for nt in 0 1 2 3; do
  export DAZIM_LIB=/tmp/libdazim_nt$nt.so
  DAZIM_HIPCC_EXTRA="-DDZ_FMM_NT=$nt" python -c "import dazimsurftomo_amd as dz; dz.build(force=True)" > /dev/null 2>&1 || echo build failed
  echo "== DZ_FMM_NT=$nt"; python tools/fmm_only.py 1000 2 2>&1 | grep -E "kernel|checksum" | tail -2
done
