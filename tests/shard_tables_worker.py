"""Worker of tests/test_rehearsal_gpu.py: `world` processes on ONE GPU over the file transport (dazim_comm_init_files).

    python tests/shard_tables_worker.py <rank> <world> <comm dir> <out.npz> <ny>

Every rank holds the same seeded model (nx = 5 columns per row, `ny` rows -- fewer rows than ranks leaves ranks without a block)
and calls the library's model-sharded table entries, which compute this rank's block of rows and join the blocks by all-gathers
inside the library: dazim_dispersion_kernels_sharded (device-resident with option disp.async: the depth-kernel tables are gathered
when the auxiliary stream is joined; host arrays: at once; curves only) and dazim_ti_kernels_sharded.  Beside them the plain
single-rank calls on the whole model.  The test wants every table of every rank equal to the plain call's, bit for bit."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def model(ny, nx=5):
    rng = np.random.default_rng(9)
    depz = np.array([0.0, 5.0, 12.0, 25.0, 45.0, 70.0], np.float32)
    periods = np.array([6.0, 10.0, 16.0, 25.0])
    vel = (3.0 + 0.2 * np.arange(len(depz))[:, None, None] + 0.05 * rng.standard_normal((len(depz), ny, nx))).astype(np.float32)
    return vel, depz, periods


def main(rank, world, comm_dir, out_path, ny):
    import torch
    import dazimsurftomo_amd as dz
    ctx = dz.Context(0)
    vel, depz, periods = model(ny)
    d_vel = torch.from_numpy(vel).to("cuda:0")
    # the single-rank tables (no communicator yet)
    pv1, sen1, nf1 = ctx.depthkernel(vel, depz, periods, 2.0)
    ls1 = ctx.ti_kernels(vel, depz, periods, 2.0, pv1)
    ctx.comm_init_files(world, rank, comm_dir)
    out = {"pv1": pv1, "ls1": ls1, "nf1": nf1}
    for q in range(3):
        out[f"sen1_{q}"] = sen1[q]
    # (a) host arrays: everything joined when the call returns
    pv, sen, nf = ctx.depthkernel(vel, depz, periods, 2.0, sharded=True)
    out["pv_host"], out["nf_host"] = pv, nf
    for q in range(3):
        out[f"sen_host_{q}"] = sen[q]
    # (b) curves only
    pvo, _, nfo = ctx.depthkernel(vel, depz, periods, 2.0, kernels=False, sharded=True)
    out["pv_only"] = pvo
    # (c) device-resident, two streams: pv complete on return, the depth kernels after the join (dazim_sync here)
    ctx.set_option("disp.async", 1)
    pvd, send, nfd = ctx.depthkernel(d_vel, depz, periods, 2.0, sharded=True)
    out["pending_after_call"] = ctx.stat("aux.pending")
    out["pv_dev"] = pvd.cpu().numpy()          # (a torch copy on torch's stream: pv was complete when the call returned)
    ctx.sync()
    out["pending_after_sync"] = ctx.stat("aux.pending")
    for q in range(3):
        out[f"sen_dev_{q}"] = send[q].cpu().numpy()
    # (c') the same again, joined by the next sharded call instead (the send buffer is reused: the join must come first)
    pvd2, send2, _ = ctx.depthkernel(d_vel, depz, periods, 2.0, sharded=True)
    pvd3, send3, _ = ctx.depthkernel(d_vel, depz, periods, 2.0, sharded=True)
    ctx.sync()
    out["sen_dev2_same"] = all(bool(torch.equal(a, b)) for a, b in zip(send2, send3))
    out["sen_dev2_0"] = send2[0].cpu().numpy()
    ctx.set_option("disp.async", 0)
    # (d) TI kernels, host and device
    out["ls_host"] = ctx.ti_kernels(vel, depz, periods, 2.0, pv1, sharded=True)
    out["ls_dev"] = ctx.ti_kernels(d_vel, depz, periods, 2.0, torch.from_numpy(pv1).to("cuda:0"), sharded=True).cpu().numpy()
    # (e) the primitive
    g = ctx.comm_allgather(np.array([rank * 10.0, rank * 10.0 + 1], np.float32))
    out["gather"] = g
    ctx.comm_free()
    ctx.close()
    np.savez(out_path, **out)


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5]))
