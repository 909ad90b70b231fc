#!/bin/bash
# eikonal kernel on all / half / a quarter of the CUs (8 workgroups per CU each): compute-bound per CU or memory-system-bound?
F=ffffffff; H=55555555; Q=11111111; Z=00000000
run() { echo "== $1"; DAZIM_CU_MASK=$2 python tools/fmm_only.py ${3:-1000} 2 2>&1 | grep kernel | tail -1; }
run "all 256 CUs" $F,$F,$F,$F,$F,$F,$F,$F
run "every other CU (128)" $H,$H,$H,$H,$H,$H,$H,$H
run "first 128 CUs" $F,$F,$F,$F,$Z,$Z,$Z,$Z
run "every 4th CU (64)" $Q,$Q,$Q,$Q,$Q,$Q,$Q,$Q
run "first 64 CUs" $F,$F,$Z,$Z,$Z,$Z,$Z,$Z
