"""The joint LSMR system of test4_Yunnan's first outer iteration (94 317 x 73 440, 53.6 M entries) built as tests/test_e2e_test4_gpu.py
builds it, then solved `reps` times:  python tools/lsmr_test4.py [reps]   (OPTS=spmv.split=0,... sets library options)
prints the solver's seconds, iterations and the product kernels' microseconds (products <= 65 us, LSMR <= 0.19 s)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dazimsurftomo_amd as dz  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402  (the regularisation rows come from the checker, as in the test)
from tests.test_e2e_test4_gpu import GOLD, GOLD_J, flatten  # noqa: E402

d, j = np.load(GOLD), np.load(GOLD_J)
ctx, orc = dz.Context(0), Oracle()
for kv in os.environ.get("OPTS", "").split(","):
    if "=" in kv:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
nx, ny, nz = int(d["nx"]), int(d["ny"]), int(d["nz"])
goxd, gozd, dv, minthk = float(d["goxd"]), float(d["gozd"]), float(d["dv"]), float(d["minthk"])
vel, depz, t = d["vel"], d["depz"], d["t"]
pv, sen, nfail = ctx.depthkernel(vel, depz, t, minthk)
lsen = ctx.ti_kernels(vel, depz, t, minthk, pv)
scx, scz, per, ray_f, rx, rz = flatten(d["scxf"], d["sczf"], d["rcxf"], d["rczf"], d["nrc1"], d["nsrc1"], d["periods"])
fields = ctx.fmm_batch(nx, ny, goxd, gozd, dv, dv, pv, scx, scz, per)
G, tpred, nb = ctx.rays_build_G(nx, ny, goxd, gozd, dv, dv, vel, fields, scx, scz, per, ray_f, rx, rz, sen, lsen=lsen)
dall, nvp = len(tpred), (nx - 2) * (ny - 2) * (nz - 1)
G.scale_rows(j["w"])
e, ei = np.zeros(0, np.float32), np.zeros(0, np.int32)
c1, rw1, ir1, ic1 = orc.tikhonov_iso(nx, ny, nz, dall, 20.0, e, ei, ei)
c2, rw2, ir2, ic2 = orc.tikhonov_iso(nx, ny, nz, dall, 30.0, e, ei, ei)
G.append_coo(3 * c1, np.concatenate([ir1, ir2 + c1, ir2 + 2 * c1]).astype(np.int32),
             np.concatenate([ic1, ic2 + nvp, ic2 + 2 * nvp]).astype(np.int32), np.concatenate([rw1, rw2, rw2]))
b = np.zeros(dall + 3 * c1, np.float32)
b[:dall] = (d["obst"] - tpred) * j["w"]
print(f"G {G.m} x {G.n}, {G.nnz} entries")
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    x, info = ctx.lsmr(G, b, 0.0, 1e-5, 1e-4, 200.0, 500, 10)
    print(f"lsmr {ctx.kernel_seconds('lsmr') * 1e3:.1f} ms  itn {info['itn']} istop {info['istop']}  A*x {ctx.kernel_seconds('spmv') * 1e6:.1f} us "
          f"(kind {ctx.kernel_seconds('spmv.kind'):.0f})  A^T*y {ctx.kernel_seconds('spmvt') * 1e6:.1f} us (kind {ctx.kernel_seconds('spmvt.kind'):.0f})  "
          f"split row {ctx.kernel_seconds('spmv.split_row'):.0f}  |x| {np.linalg.norm(x):.6f}")
G.free()
