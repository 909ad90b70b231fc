#!/bin/bash
# same-box A/B of prebuilt libraries on the eikonal kernel alone: tools/ab_libs.sh <libA.so> <libB.so> ...   ("-" = the product library)
# env SRC = sources (x 16 periods), NX = model columns per side, REPS = interleaved repetitions
for rep in $(seq 1 ${REPS:-3}); do
  for l in "$@"; do
    if [ "$l" = "-" ]; then unset DAZIM_LIB; else export DAZIM_LIB=$PWD/$l; fi
    echo -n "[$l] "; python tools/fmm_only.py ${SRC:-1000} 2 2>&1 | grep kernel | tail -1 | awk '{print $7, $8, $9, $10}'
  done
done
