DAZIM_BENCH_REHEARSAL=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29590 bench.py --gpus 2 --no-cpu --steps 1 --warmup 1 > gpurun_out/reh.out 2> gpurun_out/reh.err
echo rc $?
tail -c 1500 gpurun_out/reh.err | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl"
grep "^{" gpurun_out/reh.out | head -c 600
