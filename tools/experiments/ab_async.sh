# same-box A/B: eikonal call asynchronous with the ray kernel's count pass beside its tail (fmm.async, default in bench.py) or one after the other
for i in 1 2 3; do
for o in 0 1; do
DAZIM_FMM_ASYNC=$o python bench.py --steps 6 --warmup 2 --no-cpu "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('fmm.async=$o', round(d['ms_per_step'],2), d['phases_s'], d.get('rays_beside_eikonal_tail'))"
done; done
