#!/bin/bash
# time-sliced eikonal marches: stage-count / taper sweep on the S-256 synthetic batch (same box), checksums printed
for o in fmm.ts_stages=2 fmm.ts_stages=3 fmm.ts_stages=4 fmm.ts_stages=6 fmm.ts_stages=8 fmm.ts_stages=12 fmm.ts_stages=3,fmm.ts_taper=70 fmm.ts_stages=4,fmm.ts_taper=70 fmm.ts_stages=2,fmm.ts_taper=150 fmm.ts_stages=3,fmm.ts_taper=150 fmm.ts_stages=2 ; do
  OPTS=$o python tools/fmm_only.py ${1:-1000} 1 2>&1 | grep -E "kernel|checksum" | tr "\n" " " | awk -v o=$o '{print o, $7, $8, $9, $16, $17}'
done
