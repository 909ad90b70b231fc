# bit-for-bit comparison of a whole bench step with the eikonal call asynchronous (ray pass beside its tail) and synchronous:
#   bash tools/experiments/stress_async.sh [bench flags...]      env OPTS=... passes library options (e.g. fmm.cap=512: thousands of spill reruns)
for o in 0 1; do
DAZIM_FMM_ASYNC=$o DAZIM_OPTS="$OPTS" python bench.py --steps 1 --warmup 0 --no-cpu --dump /tmp/dump_$o "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('async=$o', round(d['ms_per_step'],1), d['phases_s'], d.get('rays_beside_eikonal_tail'))"
done
python - <<'PY'
import numpy as np
a=np.load('/tmp/dump_0.0.npz'); b=np.load('/tmp/dump_1.0.npz')
print('tpred equal', np.array_equal(a['tpred'], b['tpred']), 'x equal', np.array_equal(a['x'], b['x']), float(np.abs(a['x']).max()))
PY
