#!/bin/bash
# time-sliced marches: many hand-overs, repeated -- every run must give the checksum of the unsliced kernel
ref=$(OPTS=fmm.ts=2,fmm.hyb512=2 python tools/fmm_only.py ${1:-1000} 1 2>&1 | grep checksum)
echo "unsliced: $ref"
bad=0
for rep in 1 2 3; do
  for o in fmm.ts_stages=15 fmm.ts_stages=15,fmm.ts_taper=70 fmm.ts_stages=8,fmm.ts_taper=50 fmm.ts_stages=4,fmm.ts_taper=50 fmm.ts_stages=15,fmm.hyb512=2 fmm.ts_stages=2; do
    c=$(OPTS=fmm.ts=1,$o python tools/fmm_only.py ${1:-1000} 1 2>&1 | grep checksum)
    if [ "$c" != "$ref" ]; then echo "MISMATCH rep $rep $o: $c"; bad=$((bad+1)); fi
  done
done
echo "mismatches: $bad of 18"
