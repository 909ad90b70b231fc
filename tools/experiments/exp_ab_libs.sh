#!/bin/bash
# same-box A/B of prebuilt libraries (they travel with the snapshot): tools/exp_ab_libs.sh libA.so libB.so ...  [SRC=sources NX=nx]
for rep in 1 2 3; do
  for l in "$@"; do
    echo -n "[$l] "; DAZIM_LIB=$PWD/dazimsurftomo_amd/lib/$l python tools/fmm_only.py ${SRC:-1000} 1 2>&1 | grep kernel | awk '{print $7, $8, $9}'
  done
done
