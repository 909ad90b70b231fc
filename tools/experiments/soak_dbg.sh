# many 25-step runs of the default (asynchronous) bench step: mean step and the slowest step of each run; a stalled step shows at once
cat > /tmp/soak_line2.py <<'PY'
import json, sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
sf = d.get('slowest_forward') or {}
print(sys.argv[1], round(d['ms_per_step'], 2), 'slowest step', sf.get('step'), round(sf.get('ms', 0), 1), 'ms  fmm', round(sf.get('fmm_s', 0) * 1e3, 1), 'rays', round(sf.get('rays_s', 0) * 1e3, 1), 'passes', sf.get('passes'))
PY
for i in $(seq 1 ${1:-40}); do
  python bench.py --steps 25 --warmup 0 --no-cpu 2> /tmp/err_$i.txt | python /tmp/soak_line2.py "run $i"
done
