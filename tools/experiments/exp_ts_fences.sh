#!/bin/bash
# what the two agent-scope fences of a time-sliced hand-over cost: builds that leave out the L2 write-back of the release and / or
# the L2 invalidate of the acquire (NOT coherent across XCDs -- measurement only, checksums printed), 2 / 8 / 15 stages
i=0
for f in "" "-DDZ_TS_LIGHT_REL" "-DDZ_TS_LIGHT_ACQ" "-DDZ_TS_LIGHT_REL -DDZ_TS_LIGHT_ACQ"; do
  export DAZIM_LIB=/tmp/libdazim_tsf$i.so
  DAZIM_HIPCC_EXTRA="$f" python -c "import dazimsurftomo_amd as dz; dz.build(force=True)" > /dev/null 2>&1 || echo "build failed: $f"
  for n in 2 8 15; do
    OPTS=fmm.ts_stages=$n python tools/fmm_only.py ${1:-1000} 1 2>&1 | grep -E "kernel|checksum" | tr "\n" " " | awk -v o="[$f] stages=$n" '{print o, $7, $8, $9, $16}'
  done
  i=$((i+1))
done
