# same-box A/B: coarse fields handed to the ray kernel column-major (DAZIM_BENCH_TTN=1, rounds 1-5) or kept in the eikonal kernel's tiles
for i in 1 2 3; do
for o in 1 0; do
DAZIM_BENCH_TTN=$o python bench.py --steps 6 --warmup 2 --no-cpu "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('column-major ttn' if $o else 'tiled fields    ', round(d['ms_per_step'],2), d['phases_s'])"
done; done
