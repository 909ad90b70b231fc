"""Experiment: do the dispersion kernel (fp64 VALU) and the eikonal kernel (LDS-heap, two wavefronts per SIMD) share the chip
when they run on two streams?  Two contexts, two host threads; S-256 workload of bench.py."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import dazimsurftomo_amd as dz
from bench import *  # noqa

def main():
    dev = torch.device("cuda:0")
    class A: nsrc = 1000; nrcv = 32; scaling = "weak"
    bench.set_workload("s256")
    from bench import NX, NY, GOXD, GOZD, DV, DEPZ, PERIODS, MINTHK
    vel = bench.s256_model()
    scx, scz, per, field_of_ray, rcx, rcz = bench.workload(1000, 32, 0)[:6]
    nfield = len(scx)
    g = dz.geometry(NX, NY, GOXD, GOZD, DV, DV)
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    d_vel, d_scx, d_scz, d_per = T(vel), T(scx), T(scz), T(per)
    kmax, ncol, nz = len(PERIODS), NX * NY, len(DEPZ)
    mk = lambda: (torch.empty((kmax, ncol), dtype=torch.float64, device=dev),
                  [torch.empty((nz, kmax, ncol), dtype=torch.float64, device=dev) for _ in range(3)])
    d_pv, d_sen = mk()
    d_pv2, d_sen2 = mk()
    d_veln = torch.empty((kmax, g.nnx, g.nnz), dtype=torch.float32, device=dev)
    d_ttn = torch.empty((nfield, g.nnx, g.nnz), dtype=torch.float32, device=dev)
    d_ttnr = torch.empty((nfield, 129, 129), dtype=torch.float32, device=dev)
    d_nstsr = torch.empty((nfield, 129, 129), dtype=torch.int32, device=dev)
    d_box = torch.empty((nfield, 12), dtype=torch.int32, device=dev)
    d_st = torch.empty((nfield,), dtype=torch.int32, device=dev)
    ca, cb = dz.Context(0), dz.Context(0)
    def disp(c, pv, sen):
        return c.depthkernel(d_vel, DEPZ, PERIODS, MINTHK, pv=pv, sen=sen)
    def fmm(c):
        return c.fmm_batch(NX, NY, GOXD, GOZD, DV, DV, d_pv, d_scx, d_scz, d_per, veln=d_veln, ttn=d_ttn,
                           ttnr=d_ttnr, nstsr=d_nstsr, boxes=d_box, status=d_st)
    disp(ca, d_pv, d_sen); fmm(ca); disp(cb, d_pv2, d_sen2)
    def wall(f):
        torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); return time.perf_counter() - t
    for occ in ((3, 2, 1) if not os.environ.get("DISP_FIRST") else (3,)):
        cb.set_option("disp.occ", occ)
        td = min(wall(lambda: disp(cb, d_pv2, d_sen2)) for _ in range(2))
        tf = min(wall(lambda: fmm(ca)) for _ in range(2))
        for delay in ((0.0, 0.005) if not os.environ.get("DISP_FIRST") else (0.0, 0.02, 0.04, 0.05, 0.056, 0.062)):
            def both():
                if os.environ.get("DISP_FIRST"):   # the dispersion kernel first, the eikonal kernel `delay` later (its tail overlapped)
                    th = threading.Thread(target=lambda: disp(cb, d_pv2, d_sen2)); th.start()
                    if delay: time.sleep(delay)
                    fmm(ca); th.join()
                    return
                th = threading.Thread(target=lambda: fmm(ca)); th.start()
                if delay: time.sleep(delay)
                disp(cb, d_pv2, d_sen2); th.join()
            tb = min(wall(both) for _ in range(3))
            print(f"disp.occ={occ} delay={delay*1e3:.0f}ms  disp alone {td*1e3:.1f} ms  fmm alone {tf*1e3:.1f} ms  sum {1e3*(td+tf):.1f}  both {tb*1e3:.1f} ms "
                  f"(kernel fmm {ca.kernel_seconds('fmm')*1e3:.1f}, disp {cb.kernel_seconds('disp')*1e3:.1f})", flush=True)

main()
