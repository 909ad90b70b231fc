#!/bin/bash
# time-sliced eikonal marches: stage-count sweep on the S-256 / S-512 synthetic batches (same box)
for n in 2 3 4 6 8 12; do
  OPTS=fmm.ts=1,fmm.ts_stages=$n python tools/fmm_only.py ${1:-1000} 1 2>&1 | grep -E "kernel" | sed "s/^/768 stages=$n /"
  OPTS=fmm.ts=1,fmm.hyb512=1,fmm.ts_stages=$n python tools/fmm_only.py ${1:-1000} 1 2>&1 | grep -E "kernel" | sed "s/^/hyb512 stages=$n /"
done
for n in 4 8 16 24; do
  NX=105 OPTS=fmm.ts=1,fmm.ts_stages=$n python tools/fmm_only.py ${1:-1000} 1 2>&1 | grep -E "kernel" | sed "s/^/S-512 stages=$n /"
done
