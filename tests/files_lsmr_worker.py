"""Worker of tests/test_multirank_files_gpu.py: `world` processes on ONE GPU, the library's row-sharded LSMR over the file transport
(dazim_comm_init_files): every line of the sharded solve runs with 2 or 3 ranks where RCCL would refuse the device used twice.
    python tests/files_lsmr_worker.py <rank> <world> <comm dir> <out.json>
Each rank builds the same seeded system as tests/dist_lsmr_worker.py, keeps its row shard, solves, and writes x and the solver's
info; the test compares every rank with the single-process oracle and with each other (bit for bit)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(rank, world, comm_dir, out_path):
    import torch
    import dazimsurftomo_amd as dz
    from dazimsurftomo_amd.distributed import shard_rows
    from tests.dist_lsmr_worker import CFG, system
    ctx = dz.Context(0)
    m, n, irow, icol, rw, b = system()
    r0, r1 = shard_rows(m, world, rank)
    keep = (irow > r0) & (irow <= r1)
    G = ctx.csr_from_coo(r1 - r0, n, (irow[keep] - r0).astype(np.int32), icol[keep], rw[keep])
    b_loc = torch.from_numpy(b[r0:r1].copy()).to("cuda:0")
    ctx.comm_init_files(world, rank, comm_dir)
    # the generic reduction the sharded host program uses, on a host array and on a device array
    h = np.array([rank + 1.0, 10.0 * (rank + 1)], np.float64)
    ctx.comm_allreduce(h)
    hmax = np.array([rank, -rank], np.int64)
    ctx.comm_allreduce(hmax, "max")
    x, info = ctx.lsmr(G, b_loc, *CFG, x=torch.zeros(n, dtype=torch.float32, device="cuda:0"))
    out = {"x": x.cpu().numpy().tolist(), "info": info, "nranks": int(ctx.kernel_seconds("lsmr.nranks")),
           "transport": int(ctx.kernel_seconds("lsmr.transport")), "collectives_per_iteration": ctx.kernel_seconds("lsmr.collectives_per_iteration"),
           "sum": h.tolist(), "max": hmax.tolist()}
    ctx.comm_free()
    G.free()
    ctx.close()
    json.dump(out, open(out_path, "w"))


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4])
