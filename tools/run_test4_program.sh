#!/bin/bash
# the bundled test4_Yunnan example through host/DAzimSurfTomo_amd (inputs from tests/golden), plain and under rocprofv3
root=$PWD
d=/tmp/t4run; rm -rf $d; mkdir -p $d
python - <<PY
import numpy as np
g = np.load("$root/tests/golden/test4_yunnan_full.npz")
open("$d/para.in", "w").write(str(g["para"])); open("$d/China_YN_Rayleigh_RS_5-40s.dat", "w").write(str(g["data"])); open("$d/MOD", "w").write(str(g["mod"]))
PY
cd $d
for i in 1 2; do s=$(date +%s%N); DAZIM_TIMING=1 $root/host/DAzimSurfTomo_amd para.in 2>&1 | grep -E "All time cost|phase seconds"; e=$(date +%s%N); echo "wall $(( (e - s) / 1000000 )) ms (process start to exit)"; done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $d/prof -o t4 -- $root/host/DAzimSurfTomo_amd para.in > $d/prof.log 2>&1
cd $root
python tools/rocpd_summary.py $(find $d/prof -name "*results.db" | head -1) | head -16
