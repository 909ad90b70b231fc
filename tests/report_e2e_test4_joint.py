"""Parity + timing numbers of the joint test4_Yunnan iteration (not collected by pytest): python tests/report_e2e_test4_joint.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dazimsurftomo_amd as dz
from oracle.pyoracle import Oracle
from tests.test_rays_gpu import flatten
G0 = os.path.join(os.path.dirname(__file__), "golden")
d, j = np.load(os.path.join(G0, "test4_yunnan.npz")), np.load(os.path.join(G0, "test4_yunnan_joint.npz"))
ctx, orc = dz.Context(0), Oracle()
nx, ny, nz = int(d["nx"]), int(d["ny"]), int(d["nz"]); goxd, gozd, dv, minthk = float(d["goxd"]), float(d["gozd"]), float(d["dv"]), float(d["minthk"])
vel, depz, t = d["vel"], d["depz"], d["t"]
scx, scz, per, ray_f, rx, rz = flatten(d["scxf"], d["sczf"], d["rcxf"], d["rczf"], d["nrc1"], d["nsrc1"], d["periods"])
for rep in range(2):
    t0 = time.time(); pv, sen, nfail = ctx.depthkernel(vel, depz, t, minthk); t1 = time.time()
    lsen = ctx.ti_kernels(vel, depz, t, minthk, pv); t2 = time.time()
    fields = ctx.fmm_batch(nx, ny, goxd, gozd, dv, dv, pv, scx, scz, per); t3 = time.time()
    G, tpred, nb = ctx.rays_build_G(nx, ny, goxd, gozd, dv, dv, vel, fields, scx, scz, per, ray_f, rx, rz, sen, lsen=lsen); t4 = time.time()
    print(f"rep {rep}: depthkernel {t1-t0:.3f}s  TI {t2-t1:.3f}s (kernel {ctx.kernel_seconds('ti'):.4f})  fmm {t3-t2:.3f}s  rays+G {t4-t3:.3f}s nnz {G.nnz}")
    if rep == 0: G.free()
dl = np.abs(lsen - j["lsen"]); mx = np.abs(j["lsen"]).max()
print("Lsen: max|d|/max %.2e, within 1e-6: %.5f" % (dl.max() / mx, (dl <= 1e-6 * mx).mean()))
dall, nvp = len(tpred), (nx - 2) * (ny - 2) * (nz - 1)
G.scale_rows(j["w"])
e = np.zeros(0, np.float32); ei = np.zeros(0, np.int32)
c1, rw1, ir1, ic1 = orc.tikhonov_iso(nx, ny, nz, dall, 20.0, e, ei, ei); c2, rw2, ir2, ic2 = orc.tikhonov_iso(nx, ny, nz, dall, 30.0, e, ei, ei)
G.append_coo(3 * c1, np.concatenate([ir1, ir2 + c1, ir2 + 2 * c1]).astype(np.int32), np.concatenate([ic1, ic2 + nvp, ic2 + 2 * nvp]).astype(np.int32), np.concatenate([rw1, rw2, rw2]))
b = np.zeros(dall + 3 * c1, np.float32); b[:dall] = (d["obst"] - tpred) * j["w"]
t0 = time.time(); x, info = ctx.lsmr(G, b, 0.0, 1e-5, 1e-4, 200.0, 500, 10); t1 = time.time()
print("lsmr %.3fs (kernel %.3f; per product: A*x %.1f us, A^T*y %.1f us)" % (t1 - t0, ctx.kernel_seconds("lsmr"), ctx.kernel_seconds("spmv") * 1e6, ctx.kernel_seconds("spmvt") * 1e6), info, "ref", j["info"])
for blk, nm in enumerate(("dVs", "Gc", "Gs")):
    a, r = x[blk * nvp:(blk + 1) * nvp], j["x"][blk * nvp:(blk + 1) * nvp]
    print(nm, "rel-L2 %.2e  max|d| %.2e  max|ref| %.3f" % (np.linalg.norm(a - r) / np.linalg.norm(r), np.abs(a - r).max(), np.abs(r).max()))
