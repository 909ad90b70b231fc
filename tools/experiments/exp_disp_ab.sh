#!/bin/bash
# same-box A/B of dispersion-kernel builds: tools/exp_disp_ab.sh "<flags A>" "<flags B>" ...
i=0
for f in "$@"; do
  export DAZIM_LIB=/tmp/libdazim_dab$i.so
  DAZIM_HIPCC_EXTRA="$f" python -c "import dazimsurftomo_amd as dz; dz.build(force=True)" > /dev/null 2>&1 || echo "build failed: $f"
  i=$((i+1))
done
for rep in 1 2 3; do
  i=0
  for f in "$@"; do
    echo -n "[$f] "; DAZIM_LIB=/tmp/libdazim_dab$i.so python tools/disp_only.py 2>&1 | tail -1
    i=$((i+1))
  done
done
