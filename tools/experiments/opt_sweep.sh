#!/bin/bash
# usage: tools/opt_sweep.sh <option> <v1> <v2> ...   -- bench phase times for several values of one tuning option
opt=$1; shift
for w in "$@"; do
  DAZIM_OPTS=$opt=$w python bench.py --no-cpu --steps 1 --warmup 1 2>&1 | tail -1 > /tmp/_sweep.json
  python -c "import json; d=json.load(open('/tmp/_sweep.json')); print('$opt=$w', d['value'], d['phases_s'])"
done
