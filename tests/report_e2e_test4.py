"""Timing + parity report of the bundled test4_Yunnan example through the host-array API (not collected by pytest; run on the GPU box:
python tests/report_e2e_test4.py).  Uses the oracle for the Tikhonov rows, hence lives under tests/."""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import dazimsurftomo_amd as dz
from oracle.pyoracle import Oracle
from tests.test_rays_gpu import flatten
d = np.load("tests/golden/test4_yunnan.npz"); ctx = dz.Context(0); orc = Oracle()
nx, ny, nz = int(d["nx"]), int(d["ny"]), int(d["nz"]); goxd, gozd, dv, minthk = float(d["goxd"]), float(d["gozd"]), float(d["dv"]), float(d["minthk"])
vel, depz, t = d["vel"], d["depz"], d["t"]
for rep in range(2):
    t0 = time.time(); pv, sen, nfail = ctx.depthkernel(vel, depz, t, minthk); t1 = time.time()
    scx, scz, per, ray_f, rx, rz = flatten(d["scxf"], d["sczf"], d["rcxf"], d["rczf"], d["nrc1"], d["nsrc1"], d["periods"])
    t2 = time.time(); fields = ctx.fmm_batch(nx, ny, goxd, gozd, dv, dv, pv, scx, scz, per); t3 = time.time()
    G, tpred, nb = ctx.rays_build_G(nx, ny, goxd, gozd, dv, dv, vel, fields, scx, scz, per, ray_f, rx, rz, sen); t4 = time.time()
    print(f"rep {rep}: depthkernel {t1-t0:.3f}s (kernel {ctx.kernel_seconds('disp'):.3f}) fmm {t3-t2:.3f}s (kernel {ctx.kernel_seconds('fmm'):.3f}) rays+G {t4-t3:.3f}s (kernel {ctx.kernel_seconds('rays'):.3f})  [host-array API: PCIe included]")
    if rep == 0: G.free()
print("pv: max|d| %.2e, bit-equal fraction %.5f" % (np.abs(pv - d["pv"].astype(np.float64)).max(), (pv.astype(np.float32) == d["pv"]).mean()))
print("tpred: max rel %.2e, bit-equal fraction %.4f" % (np.abs(tpred - d["dsurf"]).max() / np.abs(d["dsurf"]).max(), (tpred == d["dsurf"]).mean()))
print("nnz gpu %d ref %d boundary rays %d" % (G.nnz, int(d["nnz"]), nb))
ir, ic, rw = G.to_coo(); dall = len(tpred); n = G.n
rs = np.bincount(ir - 1, weights=np.abs(rw).astype(np.float64), minlength=dall); cs = np.bincount(ic - 1, weights=np.abs(rw).astype(np.float64), minlength=n)
print("rowsum rel-L2 %.2e colsum rel-L2 %.2e" % (np.linalg.norm(rs - d["rowsum"]) / np.linalg.norm(d["rowsum"]), np.linalg.norm(cs - d["colsum"]) / np.linalg.norm(d["colsum"])))
e = np.zeros(0, np.float32); c3, rwT, irT, icT = orc.tikhonov_iso(nx, ny, nz, dall, 20.0, e, np.zeros(0, np.int32), np.zeros(0, np.int32))
G.append_coo(c3, irT, icT, rwT); b = np.zeros(dall + c3, np.float32); b[:dall] = d["obst"] - tpred
t0 = time.time(); x, info = ctx.lsmr(G, b, 0.0, 1e-3, 1e-3, 1200.0, 1000, 64); t1 = time.time()
print("lsmr %.3fs" % (t1 - t0), info, "ref", d["info"])
print("x rel-L2 %.2e" % (np.linalg.norm(x - d["x"]) / np.linalg.norm(d["x"])))
