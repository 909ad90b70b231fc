# with the ray kernel in the eikonal tail (fmm.async) the perturbed dispersion copies only have to be done by ~150 ms: does issue priority for the eikonal wavefronts pay now?
for i in 1 2 3; do
for o in "fmm.prio=0" "fmm.prio=1"; do
DAZIM_OPTS=$o python bench.py --steps 6 --warmup 2 --no-cpu "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$o', round(d['ms_per_step'],2), {k:round(v*1e3,1) for k,v in d['phases_s'].items()}, d['dispersion_streams'].get('perturbed_copies_s'))"
done; done
