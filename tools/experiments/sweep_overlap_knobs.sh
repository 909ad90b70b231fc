for o in "rays.sweeps=2" "rays.sweeps=1" "rays.sweeps=3" "rays.sweeps=4" "rays.sweeps=2" "rays.sweeps=1" "rays.sweeps=3" "rays.sweeps=4"; do
DAZIM_OPTS=$o python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$o', round(d['ms_per_step'],2), {k:round(v*1e3,1) for k,v in d['phases_s'].items()}, d.get('rays_passes_last_step'))"
done
