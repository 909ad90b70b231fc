#!/bin/bash
# same-box A/B of ray-kernel builds (whole bench step, phase times): tools/exp_rays_ab.sh "<flags A>" "<flags B>" ...
i=0
for f in "$@"; do
  export DAZIM_LIB=/tmp/libdazim_rab$i.so
  DAZIM_HIPCC_EXTRA="$f" python -c "import dazimsurftomo_amd as dz; dz.build(force=True)" > /dev/null 2>&1 || echo "build failed: $f"
  i=$((i+1))
done
for rep in 1 2; do
  i=0
  for f in "$@"; do
    echo -n "[$f] "; DAZIM_LIB=/tmp/libdazim_rab$i.so python bench.py --no-cpu --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['phases_s'], round(d['ms_per_step'],1))"
    i=$((i+1))
  done
done
