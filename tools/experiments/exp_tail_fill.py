"""Experiment (round 6, after round 4's exp_overlap_rays.py): does the ray kernel fill the TAIL of the eikonal launch?  Two contexts, two host threads, S-256
workload of bench.py: context A marches the batch with fmm.wg_per_cu = WPC (room left on every CU), context B traces the rays of a
previously computed copy of the fields with rays.wg_per_cu = R, started DELAY ms after the eikonal launch."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
import dazimsurftomo_amd as dz


def main():
    dev = torch.device("cuda:0")
    bench.set_workload("s256")
    from bench import NX, NY, GOXD, GOZD, DV, DEPZ, PERIODS, MINTHK
    vel = bench.s256_model()
    scx, scz, per, field_of_ray, rcx, rcz = bench.workload(1000, 32, 0)[:6]
    nfield, nray = len(scx), len(rcx)
    g = dz.geometry(NX, NY, GOXD, GOZD, DV, DV)
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    d_vel, d_scx, d_scz, d_per, d_fray, d_rcx, d_rcz = T(vel), T(scx), T(scz), T(per), T(field_of_ray), T(rcx), T(rcz)
    kmax, ncol, nz = len(PERIODS), NX * NY, len(DEPZ)
    d_pv = torch.empty((kmax, ncol), dtype=torch.float64, device=dev)
    d_sen = [torch.empty((nz, kmax, ncol), dtype=torch.float64, device=dev) for _ in range(3)]

    def bufs():
        return dict(veln=torch.empty((kmax, g.nnx, g.nnz), dtype=torch.float32, device=dev),
                    ttn=torch.empty((nfield, g.nnx, g.nnz), dtype=torch.float32, device=dev),
                    ttnr=torch.empty((nfield, 129, 129), dtype=torch.float32, device=dev),
                    nstsr=torch.empty((nfield, 129, 129), dtype=torch.int32, device=dev),
                    boxes=torch.empty((nfield, 12), dtype=torch.int32, device=dev),
                    status=torch.empty((nfield,), dtype=torch.int32, device=dev))
    ba, bb = bufs(), bufs()
    ca, cb = dz.Context(0), dz.Context(0)
    pv, sen, _ = ca.depthkernel(d_vel, DEPZ, PERIODS, MINTHK, pv=d_pv, sen=d_sen)
    d_tpred = torch.empty((nray,), dtype=torch.float32, device=dev)

    def fmm(c, b):
        return c.fmm_batch(NX, NY, GOXD, GOZD, DV, DV, pv, d_scx, d_scz, d_per, **b)
    fields_b = fmm(cb, bb)

    def rays(c, fields):
        G, tp, nb = c.rays_build_G(NX, NY, GOXD, GOZD, DV, DV, d_vel, fields, d_scx, d_scz, d_per, d_fray, d_rcx, d_rcz, sen, tpred=d_tpred)
        G.free()

    def wall(f):
        torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); return time.perf_counter() - t
    fmm(ca, ba); rays(cb, fields_b)
    tf12 = min(wall(lambda: fmm(ca, ba)) for _ in range(2))
    tr16 = min(wall(lambda: rays(cb, fields_b)) for _ in range(2))
    print(f"alone: fmm (12 wg/cu) {tf12*1e3:.1f} ms, rays (default occupancy) {tr16*1e3:.1f} ms, sum {1e3*(tf12+tr16):.1f}", flush=True)
    # round 6: NO room reserved -- the eikonal launch takes every slot (12 workgroups per CU), the ray kernel's workgroups are
    # dispatched as the eikonal ones leave: pure tail filling.  If the pair is not clearly shorter than the sum, a ray kernel that
    # waits for per-field completion flags inside one step has nothing to win.
    for delay in (0.02, 0.10, 0.15, 0.18):
        def both():
            th = threading.Thread(target=lambda: fmm(ca, ba)); th.start()
            time.sleep(delay)
            rays(cb, fields_b); th.join()
        tb = sorted(wall(both) for _ in range(4))
        print(f"full occupancy, rays launched {delay*1e3:.0f} ms after the eikonal kernel: both {tb[0]*1e3:.1f} / {tb[1]*1e3:.1f} ms against the sum {1e3*(tf12+tr16):.1f} "
              f"(kernel fmm {ca.kernel_seconds('fmm')*1e3:.1f}, rays {cb.kernel_seconds('rays')*1e3:.1f})", flush=True)

main()
