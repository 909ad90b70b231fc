for o in "fmm.ts_stages=8" "fmm.ts_stages=4" "fmm.ts_stages=2" "fmm.ts_stages=6"; do
DAZIM_OPTS=$o python bench.py --workload s512 --steps 2 --warmup 1 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$o', round(d['ms_per_step'],1), {k:round(v*1e3,1) for k,v in d['phases_s'].items()})"
done
