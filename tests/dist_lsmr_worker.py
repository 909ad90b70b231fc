"""Worker of tests/test_multigpu_gpu.py -- one process per GPU under torch.distributed.run (backend nccl = RCCL).

Every rank builds the same seeded global system, keeps its own row shard, and solves it twice: with the in-library RCCL LSMR
(dazim_comm_init + dazim_lsmr: one scalar + one n-float all-reduce per iteration on the library's stream) and with the Python
driver over torch.distributed (dazimsurftomo_amd/distributed.py).  Rank 0 writes both solutions, the per-rank agreement flags
and the rank count RCCL reports to a JSON file for the test to compare with the single-process oracle.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def system(seed=5, m=6000, n=400, per_row=12):
    """sparse least-squares system with a few dense-ish columns and 7-entry rows at the end (like G with its Tikhonov rows)"""
    rng = np.random.default_rng(seed)
    irow = np.repeat(np.arange(1, m + 1, dtype=np.int32), per_row)
    icol = rng.integers(1, n + 1, size=m * per_row).astype(np.int32)
    rw = (rng.standard_normal(m * per_row) * (1.0 + 3.0 * (icol % 17 == 0))).astype(np.float32)
    b = rng.standard_normal(m).astype(np.float32)
    return m, n, irow, icol, rw, b


CFG = (0.01, 1e-6, 1e-6, 1e8, 300, 10)   # damp, atol, btol, conlim, itnlim, localSize


def main(out_path):
    import torch
    import torch.distributed as dist
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    import dazimsurftomo_amd as dz
    from dazimsurftomo_amd.distributed import GpuLocalOps, lsmr_distributed, shard_rows
    ctx = dz.Context(local)
    m, n, irow, icol, rw, b = system()
    r0, r1 = shard_rows(m, world, rank)
    keep = (irow > r0) & (irow <= r1)
    G = ctx.csr_from_coo(r1 - r0, n, (irow[keep] - r0).astype(np.int32), icol[keep], rw[keep])
    b_loc = torch.from_numpy(b[r0:r1].copy()).to(dev)
    # (a) the Python driver over torch.distributed
    x_py, info_py = lsmr_distributed(GpuLocalOps(ctx, G), b_loc, n, *CFG)
    # (b) the in-library RCCL solve
    box = [dz.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    ctx.comm_init(world, rank, box[0])
    x_nat = torch.zeros(n, dtype=torch.float32, device=dev)
    x_nat, info_nat = ctx.lsmr(G, b_loc, *CFG, x=x_nat)
    nranks = int(ctx.kernel_seconds("lsmr.nranks"))
    torch.cuda.synchronize()
    # every rank must hold the same x, bit for bit (the state is replicated, the all-reduce gives every rank the same sums)
    same = []
    for x in (x_nat, x_py):
        parts = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(parts, x.contiguous())
        same.append(all(bool(torch.equal(parts[0], q)) for q in parts))
    # (c) the model's dispersion tables: rows sharded over the ranks + all-gather (RCCL) = the single-process call, bit for bit
    from dazimsurftomo_amd.distributed import depthkernel_sharded
    rng = np.random.default_rng(9)
    depz = np.array([0.0, 5.0, 12.0, 25.0, 45.0, 70.0], np.float32)
    periods = np.array([6.0, 10.0, 16.0, 25.0])
    vel = (3.0 + 0.2 * np.arange(len(depz))[:, None, None] + 0.05 * rng.standard_normal((len(depz), 7, 5))).astype(np.float32)
    d_vel = torch.from_numpy(vel).to(dev)
    pv1, sen1, nf1 = ctx.depthkernel(d_vel, depz, periods, 2.0)
    pv, sen, nf = depthkernel_sharded(ctx.depthkernel, d_vel, depz, periods, 2.0, world, rank, always_gather=True)
    torch.cuda.synchronize()
    disp_same = bool(torch.equal(pv, pv1)) and all(bool(torch.equal(a, b_)) for a, b_ in zip(sen, sen1)) and nf == nf1
    # ... and the product's form of it: the same sharding inside the library over ITS communicator (dazim_dispersion_kernels_sharded,
    # dazim_ti_kernels_sharded: ncclAllGather), two streams, the depth kernels gathered at the join
    ctx.set_option("disp.async", 1)
    pv_l, sen_l, nf_l = ctx.depthkernel(d_vel, depz, periods, 2.0, sharded=True)
    ctx.sync()
    ctx.set_option("disp.async", 0)
    ls1 = ctx.ti_kernels(d_vel, depz, periods, 2.0, pv1)
    ls_l = ctx.ti_kernels(d_vel, depz, periods, 2.0, pv1, sharded=True)
    torch.cuda.synchronize()
    disp_same = disp_same and bool(torch.equal(pv_l, pv1)) and all(bool(torch.equal(a, b_)) for a, b_ in zip(sen_l, sen1)) and nf_l == nf1 \
        and bool(torch.equal(ls_l, ls1))
    ctx.comm_free()
    flags = [None] * world
    dist.all_gather_object(flags, disp_same)
    infos = [None] * world
    dist.all_gather_object(infos, (info_nat, info_py))
    if rank == 0:
        json.dump({"world": world, "rccl_nranks": nranks, "x_native": x_nat.cpu().numpy().tolist(),
                   "x_python": x_py.cpu().numpy().tolist(), "info_native": info_nat, "info_python": info_py,
                   "same_x_native": same[0], "same_x_python": same[1], "disp_sharded_same": all(flags),
                   "same_info": all(i == infos[0] for i in infos)}, open(out_path, "w"))
    G.free()
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
