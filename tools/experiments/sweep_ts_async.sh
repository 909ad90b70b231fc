for o in "fmm.ts_stages=2" "fmm.ts_stages=1" "fmm.ts_stages=3" "fmm.ts_stages=4" "fmm.ts_stages=2"; do
DAZIM_OPTS=$o python bench.py --steps 5 --warmup 2 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$o', round(d['ms_per_step'],2), {k:round(v*1e3,1) for k,v in d['phases_s'].items()})"
done
