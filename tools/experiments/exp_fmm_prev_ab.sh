#!/bin/bash
# same-box A/B of the eikonal kernel against a previous fmm.hip kept as tools/experiments/fmm_prev.hip.txt (3 runs each, interleaved);
# further arguments: hipcc flag sets for more builds of the CURRENT sources:  bash tools/exp_fmm_prev_ab.sh "-DDZ_FMM_NOTAB" ...
cp tools/experiments/fmm_prev.hip.txt /tmp/fmm_prev.hip
bash tools/build_alt.sh /tmp/fmm_prev.hip /tmp/libdazim_prev.so "-I$PWD/include" > /dev/null 2>&1 || echo "alt build failed"
i=0
for f in "$@"; do
  DAZIM_LIB=/tmp/libdazim_ab$i.so DAZIM_HIPCC_EXTRA="$f" python -c "import dazimsurftomo_amd as dz; dz.build(force=True)" > /dev/null 2>&1 || echo "build failed: $f"
  i=$((i+1))
done
for rep in $(seq ${REPS:-3}); do
  echo -n "[prev] "; DAZIM_LIB=/tmp/libdazim_prev.so python tools/fmm_only.py ${SRC:-1000} 1 2>&1 | grep kernel | awk '{print $7, $8, $9}'
  echo -n "[new]  "; python tools/fmm_only.py ${SRC:-1000} 1 2>&1 | grep kernel | awk '{print $7, $8, $9}'
  i=0
  for f in "$@"; do
    echo -n "[new $f] "; DAZIM_LIB=/tmp/libdazim_ab$i.so python tools/fmm_only.py ${SRC:-1000} 1 2>&1 | grep kernel | awk '{print $7, $8, $9}'
    i=$((i+1))
  done
done
