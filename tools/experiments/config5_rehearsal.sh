#!/bin/bash
# BASELINE config 5 (S-512, 8 ranks, strong scaling) at a quarter of its source count on ONE GPU: eight ranks over the file transport
# against one rank on the same field list; tables and predicted times bit for bit, x to the LSMR bar.   bash tools/experiments/config5_rehearsal.sh [sources]
src=${1:-2000}
rm -f /tmp/c5_*.npz
python bench.py --workload s512 --scaling strong --sources $src --steps 1 --warmup 0 --no-cpu --dump /tmp/c5_one 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('1 rank :', round(d['ms_per_step'],1), 'ms', round(d['value']), 'fields/s', d['config']['workload'][:90])"
DAZIM_BENCH_REHEARSAL=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29671 bench.py --gpus 8 --workload s512 --scaling strong --sources $src --steps 1 --warmup 0 --no-cpu --dump /tmp/c5_many 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('8 ranks:', round(d['ms_per_step'],1), 'ms (one shared GPU)', round(d['value']), 'fields/s', d['lsmr']['driver'], d['lsmr']['collectives_per_iteration'], d['dispersion'][:60], 'overlap', d['rays_beside_eikonal_tail'])"
python - <<'PY'
import numpy as np
one = np.load('/tmp/c5_one.0.npz')
rs = [np.load(f'/tmp/c5_many.{r}.npz') for r in range(8)]
n = int(one['nray_all']); tp = np.full(n, np.nan, np.float32)
for r in rs:
    tp[int(r['ray0']):int(r['ray0']) + len(r['tpred'])] = r['tpred']
print('rays', n, '| tables equal', all(np.array_equal(r['pv'], one['pv']) and np.array_equal(r['sen_vs'], one['sen_vs']) for r in rs),
      '| tpred equal', np.array_equal(tp, one['tpred']), '| same x on every rank', all(np.array_equal(r['x'], rs[0]['x']) for r in rs),
      '| |x8 - x1| / |x1| = %.2e' % (np.linalg.norm(rs[0]['x'] - one['x']) / np.linalg.norm(one['x'])))
PY
