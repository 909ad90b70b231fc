for i in 1 2 3; do
for o in "ctx.pinned=0" "ctx.pinned=1"; do
DAZIM_OPTS=$o python bench.py --steps 6 --warmup 2 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$o', round(d['ms_per_step'],2), d['phases_s'])"
done; done
