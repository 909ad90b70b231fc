# soak: the last step of several 25-step asynchronous runs against one synchronous run, bit for bit (a race would show as a difference);
# per run: mean step, passes of the count kernel in the last step, the slowest step's forward part
cat > /tmp/soak_line.py <<'PY'
import json, sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(sys.argv[1], round(d['ms_per_step'], 2), 'overlap', d.get('rays_beside_eikonal_tail'), 'passes', d.get('rays_passes_last_step'), 'slowest forward', d.get('slowest_forward'))
PY
DAZIM_FMM_ASYNC=0 python bench.py --steps 2 --warmup 0 --no-cpu --dump /tmp/soak_ref "$@" > /dev/null 2>&1
for i in 1 2 3 4 5 6 7 8; do
  python bench.py --steps 25 --warmup 0 --no-cpu --dump /tmp/soak_$i "$@" 2>/dev/null | python /tmp/soak_line.py "run $i"
done
python - <<'PY'
import numpy as np
ref = np.load('/tmp/soak_ref.0.npz')
for i in range(1, 9):
    a = np.load(f'/tmp/soak_{i}.0.npz')
    print(i, 'tpred equal', np.array_equal(a['tpred'], ref['tpred']), 'x equal', np.array_equal(a['x'], ref['x']))
PY
