#!/bin/bash
# same-box A/B of eikonal-kernel builds: tools/exp_fmm_ab.sh "<flags A>" "<flags B>" ...   (each its own library, 3 runs each, interleaved)
i=0
for f in "$@"; do
  export DAZIM_LIB=/tmp/libdazim_ab$i.so
  DAZIM_HIPCC_EXTRA="$f" python -c "import dazimsurftomo_amd as dz; dz.build(force=True)" > /dev/null 2>&1 || echo "build failed: $f"
  i=$((i+1))
done
for rep in 1 2 3; do
  i=0
  for f in "$@"; do
    echo -n "[$f] "; DAZIM_LIB=/tmp/libdazim_ab$i.so python tools/fmm_only.py ${SRC:-1000} 1 2>&1 | grep kernel | awk '{print $7, $8, $9}'
    i=$((i+1))
  done
done
