#!/bin/bash
# DAzimSurfTomo_amd on N GPUs of one node, one process per GPU:   host/run_ngpu.sh N [para.in]   (from the directory with the inputs)
# Every rank reads the same inputs from its own copy of the directory (rank<r>/), takes its share of the (period, source) fields
# (host/dazim_main.f90, dazim_ranks_init) and joins one RCCL communicator whose id rank 0 leaves in comm/; the row-sharded LSMR
# runs inside the library with one collective per iteration (an all-gather, summed in rank order); each rank computes its block of the
# dispersion tables.  All ranks write the same output files; rank0/ holds them.
# DAZIM_TRANSPORT=files runs every rank on GPU 0 with the collectives staged through comm/ (tests; a one-GPU box).
set -e
n=${1:?number of GPUs}
para=${2:-para.in}
exe=$(cd "$(dirname "$0")" && pwd)/DAzimSurfTomo_amd
top=$PWD
rm -rf comm; mkdir comm
pids=()
for r in $(seq 0 $((n - 1))); do
  rm -rf rank$r; mkdir rank$r
  for f in *; do [ -f "$f" ] && cp "$f" rank$r/; done
  ( cd rank$r && DAZIM_NGPU=$n DAZIM_RANK=$r DAZIM_COMM_DIR=$top/comm DAZIM_TRANSPORT=${DAZIM_TRANSPORT:-rccl} \
      DAZIM_DEVICE=$([ "${DAZIM_TRANSPORT:-rccl}" = files ] && echo 0 || echo $r) HSA_ENABLE_IPC_MODE_LEGACY=0 \
      "$exe" "$para" > stdout.txt 2> stderr.txt ) &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=$?; done
tail -3 rank0/stdout.txt
exit $rc
