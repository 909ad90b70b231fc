for f in 4 2 1; do NX=28 OPTS=fmm.fpw=$f python tools/fmm_only.py 100 2 2>&1 | grep -E "kernel|checksum" | tail -2; done
for f in 4 2 1; do NX=28 OPTS=fmm.fpw=$f python tools/fmm_only.py 25 2 2>&1 | grep -E "kernel|checksum" | tail -2; done
