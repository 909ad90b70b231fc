"""diagnostic: the s512-sample system of tests/test_baseline_geometries_gpu.py -- products and LSMR traces, device vs oracle"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import dazimsurftomo_amd as dz
from oracle.pyoracle import Oracle
from tests.test_baseline_geometries_gpu import sample_case, oracle_rows

ctx = dz.Context(0); orc = Oracle()
vel, scx, scz, per, ray_f, rx, rz = sample_case("s512", 96, extra_long=True)
NX, NY, nz = bench.NX, bench.NY, len(bench.DEPZ)
geo = (NX, NY, bench.GOXD, bench.GOZD, bench.DV, bench.DV)
pv, sen, nfail = ctx.depthkernel(vel, bench.DEPZ, bench.PERIODS, bench.MINTHK)
g = orc.geometry(*geo)
tp_o, rw_o, ir_o, ic_o = oracle_rows(orc, g, vel, pv, sen, scx, scz, per, ray_f, rx, rz)
m, n = len(rx), (NX - 2) * (NY - 2) * (nz - 1)
rng = np.random.default_rng(5)
b0 = (rng.standard_normal(m) * 0.5).astype(np.float32)
for w in (2.0, 20.0, 200.0, 2000.0):
    c3, rwT, irT, icT = orc.tikhonov_iso(NX, NY, nz, m, w, rw_o, ir_o, ic_o)
    A = ctx.csr_from_coo(m + c3, n, irT, icT, rwT)
    b = np.zeros(m + c3, np.float32); b[:m] = b0
    for cfg in ((0.01, 1e-5, 1e-5, 1e6, 60, 10), (0.0, 1e-3, 1e-3, 1200.0, 1000, 40)):
        xd, i1 = ctx.lsmr(A, b, *cfg)
        xo, i2 = orc.lsmr(m + c3, n, irT, icT, rwT, b, *cfg)
        print("weight", w, cfg, "dev", i1["istop"], i1["itn"], round(i1["condA"], 1), "orc", i2["istop"], i2["itn"], round(i2["condA"], 1),
              "x rel", np.linalg.norm(xd - xo) / np.linalg.norm(xo), "kinds", ctx.kernel_seconds("spmv.kind"), ctx.kernel_seconds("spmvt.kind"))
    A.free()
