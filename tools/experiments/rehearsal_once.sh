#!/bin/bash
# one rehearsal of the N-rank bench on one GPU (DAZIM_BENCH_REHEARSAL, see bench.py): bash tools/experiments/rehearsal_once.sh <ranks> <bench flags...>
n=${1:-2}; shift
DAZIM_BENCH_REHEARSAL=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29590 bench.py --gpus $n --no-cpu --steps 1 --warmup 1 "$@" > gpurun_out/reh.out 2> gpurun_out/reh.err
echo rc $?
grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids\|socket.cpp\|OMP_NUM_THREADS\|\*\*\*\*" gpurun_out/reh.err | tail -5
grep "^{" gpurun_out/reh.out | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['scaling'], round(d['ms_per_step'],1), 'ms', round(d['value']), 'fields/s', d['lsmr']['driver'], d['lsmr']['rccl_nranks'], d['lsmr']['collectives_per_iteration'], d['config']['workload'][:60], d['phases_s'])"
