#!/bin/bash
# same-box A/B of library options on the whole bench step: tools/exp_opts_ab.sh "opt=1,opt2=3" "" ...
for rep in 1 2; do
  for o in "$@"; do
    echo -n "[$o] "; DAZIM_OPTS="$o" python bench.py --no-cpu --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['phases_s'], round(d['ms_per_step'],1))"
  done
done
